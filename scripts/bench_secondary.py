#!/usr/bin/env python3
"""The kernels outside the headline tick at a size that matters, for `rocprofv3 --kernel-trace --stats`
(scripts/gpu_job.sh <tag> secondary -> profiles/archive/r03_secondary_kernel_stats_*.csv) and as wall-clock rates:

  k_field_generic     16 384 chunk fields on a map whose passable cells cost 1..4 (the BFS kernel declines
                      every chunk: all requests take the LDS relaxation), device resident
  k_region_field      256 enemy-seek style jobs (128 x 128 padded region -> 64 x 64 window, ~40 seeds each):
                      one asynchronous field batch of the reference (MAX_FIELD_TASKS, nav.c:88)
  k_blockers_circles  10 000 circles (configs[4]'s obstacle count) + k_refresh_touched + k_local_islands
  k_local_islands     the relabel of every chunk of the map
  k_los_field         4 096 destination-chunk LOS fields
Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge            # noqa: E402
ge.build_navhip()
from permafrost_engine_amd import navhip, synth   # noqa: E402


def timed(fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def main():
    import torch
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    W, K = 16, 64
    rng = np.random.RandomState(5)
    grid = synth.cost_grid(W, W, seed=1234)
    out = {}

    # ---- generic field kernel: non-unit costs everywhere
    costly = grid.copy()
    passable = costly != 255
    costly[passable] = rng.randint(1, 5, size=int(passable.sum())).astype(np.uint8)
    ctx = navhip.NavContext(W, W)
    ctx.upload_plane(0, navhip.PLANE_COST_BASE, synth.to_chunks(costly))
    ctx.upload_plane(0, navhip.PLANE_BLOCKERS, np.zeros((W, W, 64, 64), np.uint16))
    ctx.relabel_local_islands(0)
    liid = synth.from_chunks(ctx.download_plane(0, navhip.PLANE_LOCAL_ISLANDS))
    dests = synth.destinations(grid, K, seed=42)
    cols = synth.whole_map_requests(grid, dests, liid)
    reqs = navhip.make_reqs(len(cols["type"]))
    for k in synth.REQ_FIELDS:
        reqs[k] = cols[k]
    d_reqs = torch.from_numpy(reqs.view(np.uint8).reshape(len(reqs), 32)).to(dev)
    pool = torch.zeros((len(reqs), 4096), dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(device=dev)

    def gen():
        ctx.build_fields_dev(d_reqs, len(reqs), pool, stream=st.cuda_stream)
        st.synchronize()
    t = timed(gen, reps=10)
    out["field_generic"] = {"chunk_fields": len(reqs), "ms": t * 1e3, "cells_per_s": len(reqs) * 4096 / t,
                            "what": "cost_base 1..4 on every passable cell: k_field_bfs declines all, k_field_generic builds all"}
    assert int((pool != 0).sum().item()) > len(reqs) * 1000
    ctx.close()

    # ---- the rest on the benchmark map
    ctx = navhip.NavContext(W, W)
    ctx.upload_plane(0, navhip.PLANE_COST_BASE, synth.to_chunks(grid))
    ctx.upload_plane(0, navhip.PLANE_BLOCKERS, np.zeros((W, W, 64, 64), np.uint16))
    t = timed(lambda: ctx.relabel_local_islands(0), reps=10)
    out["local_islands_whole_map"] = {"chunks": W * W, "ms": t * 1e3, "chunks_per_s": W * W / t}

    # region fields: one async batch of 256 jobs
    nj = 256
    rr = np.zeros(nj, navhip.REGION_REQ_DTYPE)
    seeds = []
    cells = synth.passable_cells(grid)
    for j in range(nj):
        cr, cc = rng.randint(1, W - 1), rng.randint(1, W - 1)
        rr["out_mode"][j] = 1
        rr["base_abs_r"][j], rr["base_abs_c"][j] = (cr - 1) * 64 + 32, (cc - 1) * 64 + 32
        rr["rdim"][j] = rr["cdim"][j] = 128
        rr["roff"][j] = rr["coff"][j] = 32
        rr["seed_begin"][j] = len(seeds)
        k = 40
        sr = rng.randint(rr["base_abs_r"][j], rr["base_abs_r"][j] + 128, k)
        sc = rng.randint(rr["base_abs_c"][j], rr["base_abs_c"][j] + 128, k)
        seeds += list(zip(sr, sc))
        rr["seed_count"][j] = k
    seeds = np.array(seeds, np.int16)
    t = timed(lambda: ctx.build_region_fields(rr, seeds, out_stride=4096), reps=5)
    out["region_fields_async_batch"] = {"jobs": nj, "ms_host_api": t * 1e3, "jobs_per_s": nj / t,
                                        "what": "128x128 padded regions, 64x64 window out, 40 seeds each; host buffers"}

    # blockers: 10 000 circles in, the same out
    pos = synth.cell_centre(W, W, *cells[rng.randint(len(cells), size=10_000)].T)
    circ = np.zeros(10_000, navhip.CIRCLE_DTYPE)
    circ["x"], circ["z"] = pos[:, 0], pos[:, 1]
    circ["radius"] = rng.uniform(2.0, 6.0, 10_000)
    undo = circ.copy()
    circ["delta"], undo["delta"] = 1, -1

    def blk():
        ctx.N_BlockersUpdate(circ)
        ctx.N_BlockersUpdate(undo)
    t = timed(blk, reps=5)
    out["blockers_circles"] = {"circles_per_call": 10_000, "ms_per_call_host_api": t * 1e3 / 2,
                               "circles_per_s": 20_000 / t,
                               "what": "k_blockers_circles + k_refresh_touched + k_local_islands of the touched chunks"}

    # LOS
    pick = cells[rng.randint(len(cells), size=4096)]
    lr = np.zeros(4096, navhip.LOS_REQ_DTYPE)
    lr["faction_id"] = 0xF
    lr["chunk_r"] = lr["target_chunk_r"] = pick[:, 0] // 64
    lr["chunk_c"] = lr["target_chunk_c"] = pick[:, 1] // 64
    lr["target_tile_r"], lr["target_tile_c"] = pick[:, 0] % 64, pick[:, 1] % 64
    t = timed(lambda: ctx.N_LOSFieldCreate(lr), reps=3)
    out["los_fields"] = {"fields": 4096, "ms_host_api": t * 1e3, "fields_per_s": 4096 / t}
    ctx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
