"""Build oracle/libnavoracle.so -- the plain-C restatement (oracle/navoracle.c).  TEST
INFRASTRUCTURE ONLY.  Compiled as C99 like the reference (Makefile:191 -std=c99): no FMA
contraction, so float results are reproducible against the reference's own build."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "navoracle.c")
LIB = os.path.join(HERE, "libnavoracle.so")


def build(force=False):
    deps = [SRC, os.path.join(ROOT, "include", "navhip.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    cmd = ["gcc", "-std=c99", "-D_DEFAULT_SOURCE", "-O2", "-fPIC", "-shared", "-ffp-contract=off",
           "-fno-fast-math", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"),
           SRC, "-o", LIB, "-lm", "-lpthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("gcc failed on oracle/navoracle.c")
    if r.stdout.strip():
        sys.stderr.write(r.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
