#!/usr/bin/env python3
"""Secondary measurements recorded in DESIGN.md: the host-buffer (PCIe-inclusive) rates of the two
hot entry points, LOS field throughput, repair-build throughput.  Prints one JSON object."""
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
    W, K, N = 16, 64, 100_000
    grid = synth.cost_grid(W, W, seed=1234)
    ctx = navhip.NavContext(W, W)
    ctx.upload_plane(0, navhip.PLANE_COST_BASE, synth.to_chunks(grid))
    ctx.upload_plane(0, navhip.PLANE_BLOCKERS, np.zeros((W, W, 64, 64), np.uint16))
    ctx.relabel_local_islands(0)
    liid = synth.from_chunks(ctx.download_plane(0, navhip.PLANE_LOCAL_ISLANDS))
    dests = synth.destinations(grid, K, seed=42)
    cols = synth.whole_map_requests(grid, dests, liid)
    reqs = navhip.make_reqs(len(cols["type"]))
    for k in synth.REQ_FIELDS:
        reqs[k] = cols[k]
    out = {}
    # host-buffer chunk fields: H2D requests + kernel + D2H 4 KB per field
    t = timed(lambda: ctx.N_FlowFieldUpdate(reqs), reps=3)
    out["fields_host_api_cells_per_s"] = len(reqs) * 4096 / t
    out["fields_host_api_ms"] = t * 1e3
    dirs, _ = ctx.N_FlowFieldUpdate(reqs)
    # host-buffer agent step: H2D snapshot (incl. the 67 MB field pool) + kernels + D2H results
    slot = -np.ones((K, W * W), np.int32)
    slot[cols["dest"], cols["chunk_r"] * W + cols["chunk_c"]] = np.arange(len(reqs))
    ag = synth.agents(grid, N, K, seed=7)
    offs, members = navhip.flock_csr(ag["flock"], K)
    arrays = {"pos_xz": ag["pos"], "vel_xz": ag["vel"], "radius": ag["radius"], "max_speed": ag["max_speed"],
              "speed": ag["speed"], "flags": np.full(N, navhip.ENTITY_FLAG_MOVABLE, np.uint32),
              "state": np.zeros(N, np.uint8), "has_dest_los": np.zeros(N, np.uint8), "flock": ag["flock"],
              "flock_target_xz": synth.cell_centre(W, W, dests[:, 0], dests[:, 1]),
              "flock_offsets": offs, "flock_members": members, "flock_field_slot": slot,
              "field_pool": dirs.reshape(len(dirs), 4096), "vdes_xz": None}
    t = timed(lambda: ctx.agent_step(arrays, want=("vel_xz", "new_pos_xz", "status")), reps=3)
    out["agents_host_api_steps_per_s"] = N / t
    out["agents_host_api_ms"] = t * 1e3
    arrays2 = dict(arrays)
    arrays2["field_pool"] = None
    arrays2["flock_field_slot"] = None
    arrays2["vdes_xz"] = np.tile(np.array([[1.0, 0.0]], np.float32), (N, 1))
    t = timed(lambda: ctx.agent_step(arrays2, want=("vel_xz", "new_pos_xz", "status")), reps=3)
    out["agents_host_api_given_vdes_steps_per_s"] = N / t
    # LOS: the destination-chunk field of 4096 random destinations
    rng = np.random.RandomState(1)
    cells = synth.passable_cells(grid)
    pick = cells[rng.randint(len(cells), size=4096)]
    lr = np.zeros(4096, navhip.LOS_REQ_DTYPE)
    lr["faction_id"] = 0xF
    lr["chunk_r"] = lr["target_chunk_r"] = pick[:, 0] // 64
    lr["chunk_c"] = lr["target_chunk_c"] = pick[:, 1] // 64
    lr["target_tile_r"], lr["target_tile_c"] = pick[:, 0] % 64, pick[:, 1] % 64
    t = timed(lambda: ctx.N_LOSFieldCreate(lr), reps=2)
    out["los_fields_per_s_host_api"] = 4096 / t
    out["los_ms_per_4096"] = t * 1e3
    ctx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
