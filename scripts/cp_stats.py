#!/usr/bin/env python3
"""Work counters of the ClearPath search per neighbour-count bucket (developer tool): builds a private
copy of libnavhip.so with -DNH_CP_STATS (`--build`, no GPU needed; build_prof/ travels to the GPU box)
and prints, at ticks 10 / 50 / 100 of the benchmark world, what an average problem of every bucket
costs: attempts, candidates generated / queued, cone tests, test-loop iterations, live columns."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from permafrost_engine_amd import build as nb    # noqa: E402

OUT = os.path.join(ROOT, "build_prof")
LIB = os.path.join(OUT, "libnavhip_cpstats.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for s in nb.SOURCES:
        o = os.path.join(OUT, s[:-4] + "_cpstats.o")
        subprocess.check_call([nb.HIPCC] + nb.FLAGS + ["-DNH_CP_STATS", "-c", os.path.join(nb.CSRC, s), "-o", o])
        objs.append(o)
    subprocess.check_call([nb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    print("built", LIB)


def run(crowd=0):
    os.environ["NAVHIP_LIB"] = LIB
    from permafrost_engine_amd import navhip, tick
    T = tick.NavTick(crowd_cells=crowd)
    names = ["problems", "attempts", "cand_generated", "cand_queued", "cones_left_by_compaction", "test_iters", "live_cols", "rays",
             "busy_lane_tests", "exact_iters", "outside_iters", "passes", "no_bound_after_projections", "work_cycles_columns",
             "covered_rays", "iters_projection_phase"]
    buckets = ["1-2", "3-4", "5-8", "9-16", "17-32", "33-64"]
    buf = (C.c_ulonglong * 192)()
    rows = {}
    for t in range(1, 101):
        if t in (10, 50, 100):
            T.sync()
            navhip.lib().navhip_debug_cp_work(buf, 1)
        T.step()
        if t in (10, 50, 100):
            T.sync()
            navhip.lib().navhip_debug_cp_work(buf, 1)
            out = {}
            for b, bn in enumerate(buckets):
                v = [buf[b * 16 + k] for k in range(16)]
                cy = [buf[128 + b * 8 + k] for k in range(8)]
                if v[0]:
                    out[bn] = {"problems": v[0], **{names[k]: round(v[k] / v[0], 1) for k in range(1, 16) if names[k] != "-"},
                               "kcycles_total": {"cones": cy[0] // 1000, "projections": cy[1] // 1000,
                                                 "columns": cy[2] // 1000, "jump": cy[3] // 1000, "keys_rank_compact": cy[4] // 1000},
                               "units": {"n": cy[6], "kcycles_sum": cy[5] // 1000, "kcycles_max": cy[7] // 1000}}
            rows["tick_%d" % t] = out
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        run(17 if "--crowd" in sys.argv else 0)
