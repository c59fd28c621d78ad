#!/usr/bin/env python3
"""How long does the host take to ENQUEUE one tick compared with the GPU time of the tick -- for the three drivers of
tick.NavTick: "python" (tick.py: a library call per stage), "c" (navhip_tick_run: one call per tick), "c" with
serial=True (the whole tick on one stream).  If enqueueing is not clearly faster than the tick, the GPU
waits for the host.
    python scripts/host_overhead.py [--config 0|2] [--ticks 50]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from permafrost_engine_amd import tick    # noqa: E402

DRIVERS = (("python", False), ("c", False), ("c", True))   # driver, serial

CONFIGS = {0: dict(chunk_w=4, fields_per_rank=1, agents_per_rank=1000), 2: dict(chunk_w=16, fields_per_rank=64, agents_per_rank=100_000),
           "2of8": dict(chunk_w=16, fields_per_rank=8, agents_per_rank=12_500, rank=4, world=8, shared_map=True)}


def main():
    n = int(sys.argv[sys.argv.index("--ticks") + 1]) if "--ticks" in sys.argv else 50
    which = [sys.argv[sys.argv.index("--config") + 1]] if "--config" in sys.argv else ["2", "0", "2of8"]
    for c in which:
        kw = dict(CONFIGS[int(c) if c.isdigit() else c])
        for driver, serial in DRIVERS:
            T = tick.NavTick(pipeline_fields=True, los=False, flow_velocities=not kw.get("world"), driver=driver, serial=serial, **kw)
            if kw.get("world"):
                T.pipelined, T._comm_pending = False, False          # (one rank of the job, compute only)
                # (... whose ticks write its slab only: both buffer sets start as the whole snapshot)
                T.new_pos.copy_(T.t["pos_xz"]); T.new_vel.copy_(T.t["vel_xz"])
            for _ in range(6):
                T.step()
            T.sync()
            t0 = time.perf_counter()
            for _ in range(n):
                T.step()
            t1 = time.perf_counter()
            T.sync()
            t2 = time.perf_counter()
            print("config %-4s driver %-44s enqueue %.4f ms/tick   total %.4f ms/tick" %
                  (c, T.tick_driver, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
            T.close()


if __name__ == "__main__":
    main()
