#!/usr/bin/env python3
"""A/B kernel variants inside one GPU session (box-to-box noise is ~3 %, more than most single
optimisations).  `--build NAME -DFLAG ...` cross-compiles a variant of libnavhip.so with extra flags
into build_prof/libnavhip_NAME.so (no GPU needed; the directory travels to the GPU box);
`--run A B ...` alternates bench.py over the variants (`base` = the in-tree build), 3 rounds."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from permafrost_engine_amd import build as nb    # noqa: E402

OUT = os.path.join(ROOT, "build_prof")


def build(name, flags):
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for s in nb.SOURCES:
        o = os.path.join(OUT, "%s_%s.o" % (s[:-4], name))
        subprocess.check_call([nb.HIPCC] + nb.FLAGS + flags + ["-c", os.path.join(nb.CSRC, s), "-o", o])
        objs.append(o)
    lib = os.path.join(OUT, "libnavhip_%s.so" % name)
    subprocess.check_call([nb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    for o in objs:
        os.remove(o)
    print(lib)


def run(names, rounds=3, extra=()):
    res = {n: [] for n in names}
    for _ in range(rounds):
        for n in names:
            env = dict(os.environ)
            if n != "base":
                env["NAVHIP_LIB"] = os.path.join(OUT, "libnavhip_%s.so" % n)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(extra), env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            res[n].append(d["ms_per_step"])
            print(n, round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["phase_ms"].items()}, flush=True)
    for n in names:
        v = sorted(res[n])
        print("%-12s median %.4f ms/tick  (min %.4f)" % (n, v[len(v) // 2], v[0]))


if __name__ == "__main__":
    if sys.argv[1] == "--build":
        build(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "--run":
        names = [a for a in sys.argv[2:] if not a.startswith("--rounds=")]
        rounds = [int(a.split("=")[1]) for a in sys.argv[2:] if a.startswith("--rounds=")]
        run(names, rounds=rounds[0] if rounds else 3)
