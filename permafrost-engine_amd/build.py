"""Build libnavhip.so (hand-written HIP for gfx950 + the C ABI of include/navhip.h) in-tree.

hipcc cross-compiles gfx950 without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnavhip.so")
SOURCES = ["navhip_api.hip", "pool_api.hip", "field_kernels.hip", "agent_kernels.hip", "blocker_kernels.hip", "los_kernels.hip", "region_kernels.hip", "comm_api.hip", "state_kernels.hip", "tick_api.hip", "stream_set.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         # the agent kernels mirror the reference's C arithmetic operation by operation:
         # no FMA contraction, IEEE division/sqrt, denormals kept
         "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-gpu-flush-denormals-to-zero",
         # the SLP vectoriser pairs scalar f32 chains into v_pk_* instructions whose operands must sit in aligned
         # register pairs: the moves that assemble and split the pairs cost more than the packing saves (measured,
         # profiles/r05_ab_compiler_flags.txt: tick 0.302 -> 0.296 ms, crowded world 4.78 -> 4.69).  Where packed math
         # pays (k_cohesion's distance -> weight part) the source says so itself with two-element vector types.
         "-fno-slp-vectorize",
         "-Wno-unused-result", "-Wno-unused-value", "-Wno-pass-failed", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _newer(srcs, dst):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, "navhip_internal.h"), os.path.join(ROOT, "include", "navhip.h"),
                   os.path.abspath(__file__)]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".map"))]
    if not force and not _newer(deps, LIB):
        return LIB
    objs = []
    for s in srcs:
        o = s[:-4] + ".o"
        if force or _newer(deps, o):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stdout)
                raise RuntimeError("hipcc failed on " + s)
            if verbose and r.stdout.strip():
                print(r.stdout)
        objs.append(o)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs \
        + ["-Wl,--version-script=" + os.path.join(CSRC, "navhip.map"), "-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("link of libnavhip.so failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
