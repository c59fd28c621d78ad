#!/usr/bin/env python3
"""Where does k_agent_step spend its time?  Builds a private copy of libnavhip.so with
-DNH_SECTION_PROF (s_memtime deltas per section, summed per wave with atomics), runs the bench
workload for a few ticks and prints the share of every section.  GPU box only; developer tool."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from permafrost_engine_amd import build as nb, navhip    # noqa: E402

NAMES = {0: "prologue", 1: "sp_query r30", 2: "derive r10 + filter", 3: "separation", 4: "steer/vpref",
         5: "filter + classify", 6: "clearpath"}


def main():
    out = tempfile.mkdtemp(prefix="navhip_prof_")
    objs = []
    for s in nb.SOURCES:
        o = os.path.join(out, s[:-4] + ".o")
        subprocess.check_call([nb.HIPCC] + nb.FLAGS + ["-DNH_SECTION_PROF", "-c", os.path.join(nb.CSRC, s), "-o", o])
        objs.append(o)
    lib = os.path.join(out, "libnavhip_prof.so")
    subprocess.check_call([nb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    navhip.LIB_PATH = lib
    from permafrost_engine_amd import tick
    T = tick.NavTick()
    L = navhip.lib()
    buf = (C.c_ulonglong * 32)()
    res = []
    for phase, n in (("ticks 0-4", 5), ("ticks 5-29", 25), ("ticks 30-54", 25)):
        T.sync()
        L.navhip_debug_sections(buf, 1)
        for _ in range(n):
            T.step()
        T.sync()
        L.navhip_debug_sections(buf, 1)
        v = list(buf)
        tot = sum(v[k] for k in range(7))
        row = {"phase": phase, "sections": {}}
        for k, name in NAMES.items():
            row["sections"][name] = {"share": round(v[k] / max(tot, 1), 4), "waves": v[16 + k],
                                     "ticks_per_wave": round(v[k] / max(v[16 + k], 1), 1)}
        res.append(row)
        print(phase)
        for name, d in row["sections"].items():
            print("   %-24s %6.1f%%  waves %9d  %8.1f clk/wave" % (name, 100 * d["share"], d["waves"], d["ticks_per_wave"]))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
