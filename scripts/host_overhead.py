#!/usr/bin/env python3
"""How long does the host take to ENQUEUE one tick (Python + ctypes + HIP launches) compared with the
GPU time of the tick?  If enqueueing is not clearly faster, the GPU waits for the host."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from permafrost_engine_amd import tick    # noqa: E402


def main():
    T = tick.NavTick()
    for _ in range(5):
        T.step()
    T.sync()
    for rec in (False, True):
        T.record = rec
        t0 = time.perf_counter()
        for _ in range(50):
            T.step()
        t1 = time.perf_counter()
        T.sync()
        t2 = time.perf_counter()
        print("record=%s  enqueue %.3f ms/tick   total %.3f ms/tick" % (rec, (t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
    import cProfile, pstats
    T.record = False
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        T.step()
    pr.disable()
    T.sync()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)


if __name__ == "__main__":
    main()
