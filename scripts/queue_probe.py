#!/usr/bin/env python3
"""W7: the tick of one and the same world costs X or 2X depending on what the process did before.
Hypothesis: HIP maps streams onto at most GPU_MAX_HW_QUEUES (4) hardware queues PER PRIORITY; torch's stream pool
holds 32 streams per priority, the library's side streams are pooled ones too, and whenever the caller's stream and a
side stream land on the same hardware queue their kernels serialise.  This probe builds the same world `--reps` times
in ONE process -- every NavTick takes the next stream of torch's pool -- and prints the tick time of each.
    python scripts/queue_probe.py [--config 2of8|0|2] [--reps 6] [--ticks 40]
Environment it is meant to be run under (scripts/gpu_queue_probe.sh): NAVHIP_AUX_DEDICATED, GPU_MAX_HW_QUEUES."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from permafrost_engine_amd import tick    # noqa: E402

CONFIGS = {"0": dict(chunk_w=4, fields_per_rank=1, agents_per_rank=1000),
           "2": dict(chunk_w=16, fields_per_rank=64, agents_per_rank=100_000),
           "2of8": dict(chunk_w=16, fields_per_rank=8, agents_per_rank=12_500, rank=4, world=8, shared_map=True)}


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def main():
    cfg = arg("--config", "2of8")
    reps, n = int(arg("--reps", "6")), int(arg("--ticks", "40"))
    driver = arg("--driver", "c")
    kw = dict(CONFIGS[cfg])
    out = []
    for r in range(reps):
        T = tick.NavTick(pipeline_fields=True, los=False, flow_velocities=not kw.get("world"), driver=driver,
                         time_fields=os.environ.get("PROBE_TIME_FIELDS") == "1", **kw)
        if kw.get("world"):
            T.pipelined, T._comm_pending = False, False
            T.new_pos.copy_(T.t["pos_xz"]); T.new_vel.copy_(T.t["vel_xz"])
        for _ in range(6):
            T.step()
        T.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            T.step()
        T.sync()
        dt = (time.perf_counter() - t0) / n * 1e3
        out.append(dt)
        enq = T._ctick.info().host_enqueue_ms / max(1, T._ctick.info().ticks) if T._ctick is not None else float("nan")
        print("config %-4s driver %s rep %d stream 0x%x: %.4f ms/tick (host enqueue %.4f)" % (cfg, driver, r, T.stream.cuda_stream, dt, enq), flush=True)
        T.close()
    print("config %-4s driver %s env[AUX_DEDICATED=%s MAX_HW_QUEUES=%s]: min %.4f max %.4f spread %.2fx" %
          (cfg, driver, os.environ.get("NAVHIP_AUX_DEDICATED", "-"), os.environ.get("GPU_MAX_HW_QUEUES", "-"),
           min(out), max(out), max(out) / min(out)), flush=True)


if __name__ == "__main__":
    main()
