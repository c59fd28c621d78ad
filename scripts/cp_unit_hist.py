#!/usr/bin/env python3
"""Unit durations inside k_cp (developer tool): builds a private libnavhip.so with -DNH_CP_UNIT_HIST
(`--build`) and prints, for single ticks of the benchmark world, the histogram of unit durations and how
busy the waves of the launch were (sum of wave lifetimes / (waves x kernel span))."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from permafrost_engine_amd import build as nb    # noqa: E402

OUT = os.path.join(ROOT, "build_prof")
LIB = os.path.join(OUT, "libnavhip_cphist.so")


def build(extra=()):
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for s in nb.SOURCES:
        o = os.path.join(OUT, s[:-4] + "_cphist.o")
        subprocess.check_call([nb.HIPCC] + nb.FLAGS + ["-DNH_CP_UNIT_HIST"] + list(extra) + ["-c", os.path.join(nb.CSRC, s), "-o", o])
        objs.append(o)
    subprocess.check_call([nb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    print("built", LIB)


def run(crowd=0):
    os.environ["NAVHIP_LIB"] = LIB
    from permafrost_engine_amd import navhip, tick
    T = tick.NavTick(crowd_cells=crowd)
    buf = (C.c_ulonglong * 128)()
    rows = {}
    marks = (5, 20, 40) if crowd else (5, 50, 100)
    for t in range(1, marks[-1] + 1):
        if t in marks:
            T.sync()
            navhip.lib().navhip_debug_cp_hist(buf)
        T.step()
        if t in marks:
            T.sync()
            navhip.lib().navhip_debug_cp_hist(buf)
            v = list(buf)
            rows["tick_%d" % t] = {
                "workgroup_problems_log2_ticks": {str(k): v[k] for k in range(32) if v[k]},
                "row_units_log2_ticks": {str(k): v[64 + k] for k in range(32) if v[64 + k]},
                "kernels": {name: {"longest_wave_ticks": v[96 + 8 * k + 4], "waves": v[96 + 8 * k + 3],
                                   "busy_frac": v[96 + 8 * k + 2] / max(1, v[96 + 8 * k + 3] * v[96 + 8 * k + 4])}
                            for k, name in ((0, "workgroup_problems"), (2, "rows"))}}
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build([a for a in sys.argv[2:]])
    else:
        run(17 if "--crowd" in sys.argv else 0)
