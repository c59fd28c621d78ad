#!/usr/bin/env python3
"""Generate permafrost-engine_amd/data/requests_*.npz: the chunk-field request stream of BASELINE.json's
configs as THE REFERENCE'S OWN PLANNER emits it (n_request_path, nav.c:1774 -> AStar_PortalGraphPath,
a_star.c:429), through oracle/_ref (the reference's nav.c / a_star.c / fieldcache.c compiled in place).

SURVEY.md section 8(d): "each destination expands to all chunks of the map (TARGET_TILE in the destination
chunk, TARGET_PORTAL elsewhere, portal chosen by the reference planner from the chunk centre to the
destination)".  For every (destination, chunk): one n_request_path call from the passable cell nearest to
the chunk's centre; the N_FlowFieldUpdate call the planner makes for THAT chunk is the request (its portal,
its island ids).  Chunks from which the destination cannot be reached have no request.

bench.py / tick.NavTick load the fixture for the single-GPU configs when it matches their map and
destinations (a hash of both is stored), and fall back to the numpy request generator otherwise (multi-GPU
region worlds).  Run in the build container:   python tests/tools/make_requests.py [0 1 2 3]"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pfref                                  # noqa: E402
from permafrost_engine_amd import synth                   # noqa: E402

OUT = os.path.join(ROOT, "permafrost-engine_amd", "data")
CONFIGS = {0: (4, 1), 1: (16, 16), 2: (16, 64), 3: (32, 128)}     # map side in chunks, flow fields


def world_key(grid, dests):
    h = hashlib.sha1()
    h.update(np.ascontiguousarray(grid).tobytes())
    h.update(np.ascontiguousarray(dests, np.int64).tobytes())
    return h.hexdigest()


def nearest_passable_to_centre(grid, cr, cc):
    sub = grid[cr * 64:(cr + 1) * 64, cc * 64:(cc + 1) * 64]
    ok = np.argwhere(sub != 255)
    if len(ok) == 0:
        return None
    d = np.abs(ok[:, 0] - 31.5) + np.abs(ok[:, 1] - 31.5)
    r, c = ok[int(np.argmin(d))]
    return cr * 64 + int(r), cc * 64 + int(c)


def make(cfg):
    W, K = CONFIGS[cfg]
    grid = synth.cost_grid(W, W, seed=1234)
    dests = synth.destinations(grid, K, seed=42)
    nav = pfref.RefNav(synth.to_chunks(grid))
    t0 = time.time()
    names = [n for n in pfref.FIELD_REQ_DTYPE.names]
    rows, dest_of = [], []
    for di, (R, C) in enumerate(dests):
        dst = synth.cell_centre(W, W, R, C)
        for cr in range(W):
            for cc in range(W):
                src_cell = nearest_passable_to_centre(grid, cr, cc)
                if src_cell is None:
                    continue
                if (cr, cc) == (R // 64, C // 64):
                    src_cell = (int(R), int(C))      # (in the destination chunk: from the destination itself)
                src = synth.cell_centre(W, W, src_cell[0], src_cell[1])
                ok, _ = nav.request_path(src, dst, clear_cache=True)
                reqs, _, _ = nav.trace()
                if not ok:
                    continue
                mine = [r for r in reqs if int(r["chunk_r"]) == cr and int(r["chunk_c"]) == cc]
                if not mine:
                    continue
                rows.append(mine[-1].copy())         # (a path that re-enters the chunk updates the field: the last call)
                dest_of.append(di)
        print("config %d: destination %d / %d, %d requests, %.0f s" % (cfg, di + 1, K, len(rows), time.time() - t0), flush=True)
    reqs = np.array(rows, dtype=pfref.FIELD_REQ_DTYPE)
    cols = {n: reqs[n].astype(np.int32) for n in names}
    np.savez_compressed(os.path.join(OUT, "requests_cfg%d.npz" % cfg), key=world_key(grid, dests),
                        dest=np.array(dest_of, np.int32), **cols)
    print("config %d: %d requests (%d portal) for %d destinations x %d chunks" % (
        cfg, len(reqs), int((reqs["type"] == 0).sum()), K, W * W))


def make_los(cfg):
    """The LOS-field chain of the same world: n_request_path builds, next to every flow field of a path, the LOS
    field of that chunk FROM the LOS field of the chunk before it on the path (nav.c:1843, :2026-2039), once per
    (destination, chunk): which neighbour a chunk's field is propagated from depends on the order the paths were
    requested in.  Here: per destination, one n_request_path from every chunk (row-major, the field cache kept),
    and the N_LOSFieldCreate calls it made, in order -- (chunk, previous chunk) pairs a device can replay level by
    level (navhip_build_los)."""
    W, K = CONFIGS[cfg]
    grid = synth.cost_grid(W, W, seed=1234)
    dests = synth.destinations(grid, K, seed=42)
    nav = pfref.RefNav(synth.to_chunks(grid))
    t0 = time.time()
    rows = []
    for di, (R, C) in enumerate(dests):
        dst = synth.cell_centre(W, W, R, C)
        nav.cache_clear()
        pfref.RefNav.los_trace()
        for cr in range(W):
            for cc in range(W):
                src_cell = nearest_passable_to_centre(grid, cr, cc)
                if src_cell is None:
                    continue
                if (cr, cc) == (R // 64, C // 64):
                    src_cell = (int(R), int(C))
                nav.request_path(synth.cell_centre(W, W, src_cell[0], src_cell[1]), dst, clear_cache=False)
        nav.trace()
        for did, r, c, has_prev, pr, pc in pfref.RefNav.los_trace():
            rows.append((di, r, c, (pr - r) if has_prev else 0, (pc - c) if has_prev else 0))
        if di % 8 == 7 or di == K - 1:
            print("config %d LOS: destination %d / %d, %d fields, %.0f s" % (cfg, di + 1, K, len(rows), time.time() - t0), flush=True)
    a = np.array(rows, np.int32)
    np.savez_compressed(os.path.join(OUT, "los_cfg%d.npz" % cfg), key=world_key(grid, dests), dest=a[:, 0],
                        chunk_r=a[:, 1], chunk_c=a[:, 2], prev_dr=a[:, 3], prev_dc=a[:, 4])
    print("config %d: %d LOS fields for %d destinations x %d chunks" % (cfg, len(a), K, W * W))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only_los = "--los" in sys.argv
    for c in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [0, 1, 2]:
        if not only_los:
            make(c)
        make_los(c)
