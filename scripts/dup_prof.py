#!/usr/bin/env python3
"""Cost of each section of k_agent_step under real contention: private copies of libnavhip.so run
one section twice (-DNH_DUP=k, identical results); the tick-time difference to the plain build is
that section's cost.  `--build` cross-compiles the variants into build_prof/ (no GPU needed; the
directory travels to the GPU box); without it the variants are timed.  Developer tool."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from permafrost_engine_amd import build as nb    # noqa: E402

OUT = os.path.join(ROOT, "build_prof")
VARIANTS = {0: "plain", 1: "sp_query r30 x2", 2: "derive r10 + filter x2", 3: "separation x2",
            4: "filter + classify x2", 5: "clearpath x2"}


def build():
    os.makedirs(OUT, exist_ok=True)
    common = []
    for s in nb.SOURCES:
        if s == "agent_kernels.hip":
            continue
        o = os.path.join(OUT, s[:-4] + ".o")
        subprocess.check_call([nb.HIPCC] + nb.FLAGS + ["-c", os.path.join(nb.CSRC, s), "-o", o])
        common.append(o)
    for k in VARIANTS:
        o = os.path.join(OUT, "agent_kernels_dup%d.o" % k)
        subprocess.check_call([nb.HIPCC] + nb.FLAGS + ["-DNH_DUP=%d" % k, "-c",
                               os.path.join(nb.CSRC, "agent_kernels.hip"), "-o", o])
        subprocess.check_call([nb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, "libnavhip_dup%d.so" % k), o] + common)
        print("built variant", k)


def run_one(k):
    from permafrost_engine_amd import navhip
    navhip.LIB_PATH = os.path.join(OUT, "libnavhip_dup%d.so" % k)
    from permafrost_engine_amd import tick
    T = tick.NavTick()
    for _ in range(5):
        T.step()
    T.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        T.step()
    T.sync()
    print(json.dumps({"variant": k, "name": VARIANTS[k], "ms_per_tick": (time.perf_counter() - t0) / 50 * 1e3}))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    elif "--one" in sys.argv:
        run_one(int(sys.argv[sys.argv.index("--one") + 1]))
    else:
        rows = []
        for k in VARIANTS:
            r = subprocess.run([sys.executable, __file__, "--one", str(k)], stdout=subprocess.PIPE, text=True)
            rows.append(json.loads(r.stdout.strip().splitlines()[-1]))
        base = rows[0]["ms_per_tick"]
        for r in rows:
            print("%-28s %.4f ms/tick  (+%.4f)" % (r["name"], r["ms_per_tick"], r["ms_per_tick"] - base))
