#!/usr/bin/env python3
"""Per-rank compute cost of the weak-scaling world on ONE GPU: build rank r of a `world`-rank job
and time its share of the tick (field builds + slab step, no exchange; the snapshot is not advanced,
so every tick is the same work).  Shows what the parts replicated on every rank (map planes, entity
snapshot, hash grid, flock tables) cost as the job grows.
    python scripts/rank_cost_probe.py 1 8
    python scripts/rank_cost_probe.py --strong 1 2 4 8     (ONE configs[2] world split over the ranks: bench.py
                                                            --scaling strong; compute only, no exchange)"""
import sys
import time
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from permafrost_engine_amd import tick    # noqa: E402


def main():
    strong = "--strong" in sys.argv
    drv = dict(driver="python") if "--python-driver" in sys.argv else \
        dict(driver="c", serial=False if "--no-serial" in sys.argv else None)
    base = None
    for world in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [1, 8]:
        t0 = time.time()
        if strong:
            T = tick.NavTick(rank=world // 2, world=world, shared_map=True, fields_per_rank=64 // world,
                             agents_per_rank=100_000 // world, pipeline_fields=True, **drv)
        else:
            T = tick.NavTick(rank=world // 2, world=world, **drv)
        T._comm_pending = False
        T.pipelined = False          # (no process group here: the exchange is left out)
        setup = time.time() - t0
        # (compute only: with the Python driver the snapshot is not advanced; the C driver's tick always ping-pongs its
        # buffers -- the rows of the other ranks stay what they were)
        one = T.compute if T.driver == "python" else T.step
        T.new_pos.copy_(T.t["pos_xz"]); T.new_vel.copy_(T.t["vel_xz"])     # (both buffer sets: the whole snapshot)
        for _ in range(4):
            one()
        T.sync()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            one()
        T.sync()
        dt = (time.perf_counter() - t0) / n
        base = base or dt * world
        print("[%s] world %d rank %d: setup %.1f s, %d local requests, slab %d agents of %d: %.4f ms per tick "
              "(compute only)%s" % (T.tick_driver, world, T.rank, setup, T.n_req_local, T.a1 - T.a0, T.N, dt * 1e3,
                                    "; ideal 1/%d of one rank's tick = %.4f ms: efficiency %.2f" % (world, base / world * 1e3, base / world / dt)
                                    if strong else ""), flush=True)
        T.close()


if __name__ == "__main__":
    main()
