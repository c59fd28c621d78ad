#!/usr/bin/env python3
"""A/B kernel variants inside one GPU session (box-to-box noise is ~3 %, more than most single
optimisations).  `--build NAME -DFLAG ...` cross-compiles a variant of libnavhip.so with extra flags
into build_prof/libnavhip_NAME.so (no GPU needed; the directory travels to the GPU box);
`--build-rev NAME REV` does the same for the kernel sources of a git revision (e.g. HEAD, to judge the
uncommitted change); `--run A B ... [--rounds=N] [--steps=K] [--crowded]` alternates bench.py over the
variants (`base` = the in-tree build), 3 rounds."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from permafrost_engine_amd import build as nb    # noqa: E402

OUT = os.path.join(ROOT, "build_prof")


def build(name, flags, csrc=None):
    os.makedirs(OUT, exist_ok=True)
    objs = []
    csrc = csrc or nb.CSRC
    base_flags = [f if f != "-I" + nb.CSRC else "-I" + csrc for f in nb.FLAGS]
    for s in nb.SOURCES:
        o = os.path.join(OUT, "%s_%s.o" % (s[:-4], name))
        subprocess.check_call([nb.HIPCC] + base_flags + flags + ["-c", os.path.join(csrc, s), "-o", o])
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
            lib = n
            if lib != "base":
                env["NAVHIP_LIB"] = os.path.join(OUT, "libnavhip_%s.so" % lib)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(extra), env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            res[n].append(d["ms_per_step"])
            print(n, round(d["ms_per_step"], 4), round((d.get("summary") or d)["ms_per_step_median"], 4), [round(x, 3) for x in (d.get("summary") or d).get("ms_tick_5_50_100") or [] if x is not None], flush=True)
    for n in names:
        v = sorted(res[n])
        print("%-12s median %.4f ms/tick  (min %.4f)" % (n, v[len(v) // 2], v[0]))


if __name__ == "__main__":
    if sys.argv[1] == "--build":
        build(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "--build-rev":
        src = os.path.join(OUT, "src_" + sys.argv[2])
        os.makedirs(src, exist_ok=True)
        rel = os.path.relpath(nb.CSRC, ROOT)
        for f in subprocess.check_output(["git", "ls-files", rel], cwd=ROOT, text=True).split():
            open(os.path.join(src, os.path.basename(f)), "wb").write(
                subprocess.check_output(["git", "show", "%s:%s" % (sys.argv[3], f)], cwd=ROOT))
        build(sys.argv[2], [], csrc=src)
    elif sys.argv[1] == "--run":
        names = [a for a in sys.argv[2:] if not a.startswith("--")]
        rounds = [int(a.split("=")[1]) for a in sys.argv[2:] if a.startswith("--rounds=")]
        steps = [a.split("=")[1] for a in sys.argv[2:] if a.startswith("--steps=")]
        extra = ["--no-cpu-baseline", "--no-crowded"]
        if "--crowded" in sys.argv:          # the crowded world as the main run
            extra = ["--no-cpu-baseline", "--crowded", "--warmup", "3"]
        if steps:
            extra += ["--steps", steps[0]]
        run(names, rounds=rounds[0] if rounds else 3, extra=extra)
