"""Exhaustive sweeps of the exact-arithmetic building blocks (csrc/agent_math.h) -- TEST INFRASTRUCTURE.

  build()       tests/tools/_mathsweep.so       (hipcc, gfx950: the device side, tests/tools/mathsweep.hip)
                tests/tools/_mathsweep_host.so  (gcc: what the reference computes, tests/tools/mathsweep_host.c)
                tests/tools/_mathsweep_emul.so  (g++ -DNH_HOSTSIM: agent_math.h's own algorithms with IEEE host
                                                 arithmetic -- the CPU suite's view of exp_f32_magic / cohesion_t_*)
  SWEEPS        name -> (which, [(lo_bits, hi_bits) ...]): every float of the function's domain
Built by __graft_entry__.build() (the .so files travel to the GPU box); used by tests/test_mathsweep_*.py.
"""
import ctypes
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "permafrost-engine_amd", "csrc")
DEV_SO = os.path.join(HERE, "_mathsweep.so")
HOST_SO = os.path.join(HERE, "_mathsweep_host.so")
EMUL_SO = os.path.join(HERE, "_mathsweep_emul.so")

MS_EXP, MS_SQRT_RN, MS_COH_T_F32, MS_COH_T_F64, MS_RSQ_ULP, MS_SQRT_ULP, MS_VLEN, MS_FDIV = range(8)
CHUNK_LOG2 = 22


def bits(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


# every float of the domain, as half-open ranges of bit patterns (negative floats ascend in magnitude)
SWEEPS = {
    # exp_f32_magic: a in [-104.5, -0] and [0, 89]; below -104 the argument is clamped (-> +0 like libm), the
    # cohesion weight's argument is <= 4.5 and the separation weight's <= 40
    "exp": (MS_EXP, [(0x80000000, bits(-104.5) + 1), (0, bits(89.0) + 1)]),
    # sqrt_rn_normal: s = 0 and every s in [2^-90, 2^90] (vlen's guard sends the rest to the IEEE expansion)
    "sqrt_rn": (MS_SQRT_RN, [(0, 1), (bits(2.0 ** -90), bits(2.0 ** 90) + 1)]),
    # cohesion_t_f32: len in [16, 8192); cohesion_t_f64: len in [0, 16)
    "coh_t_f32": (MS_COH_T_F32, [(bits(16.0), bits(8192.0))]),
    "coh_t_f64": (MS_COH_T_F64, [(0, bits(16.0))]),
    # vlen of (a, 0): a^2 through the guard, incl. the cold IEEE branch for tiny / huge / denormal squares
    "vlen": (MS_VLEN, [(0, bits(3.0e38) + 1)]),
    # nh_fdiv (the compiler's IEEE division, denormals kept): every finite dividend of either sign, the divisor a
    # hash of its bits (every sign and exponent, denormals included)
    "fdiv": (MS_FDIV, [(0, 0x7f800000), (0x80000000, 0xff800000)]),
}
# native approximations behind margins: largest error in ulps over the normal range the callers admit
ULP_SWEEPS = {
    "v_rsq_f32": (MS_RSQ_ULP, [(bits(2.0 ** -90), bits(2.0 ** 90) + 1)]),
    "v_sqrt_f32": (MS_SQRT_ULP, [(bits(2.0 ** -90), bits(2.0 ** 90) + 1)]),
}


def _newer(srcs, dst):
    return not os.path.exists(dst) or any(os.path.getmtime(s) > os.path.getmtime(dst) for s in srcs)


def build(force=False):
    """Compile whatever the toolchain here can (hipcc cross-compiles gfx950 without a GPU)."""
    import importlib
    import shutil
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    nb = importlib.import_module("permafrost_engine_amd.build")
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    out = {}
    src = os.path.join(HERE, "mathsweep.hip")
    if os.path.exists(nb.HIPCC) and (force or _newer([src] + hdrs, DEV_SO)):
        # the product's own arithmetic flags: no contraction, IEEE divide / sqrt, denormals kept
        r = subprocess.run([nb.HIPCC] + nb.FLAGS + ["-shared", src, "-o", DEV_SO], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on mathsweep.hip:\n" + r.stdout)
    out["device"] = DEV_SO if os.path.exists(DEV_SO) else None
    src = os.path.join(HERE, "mathsweep_host.c")
    if shutil.which("gcc") and (force or _newer([src], HOST_SO)):
        r = subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", HOST_SO, "-lm", "-lpthread"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("gcc failed on mathsweep_host.c:\n" + r.stdout)
    out["host"] = HOST_SO if os.path.exists(HOST_SO) else None
    src = os.path.join(HERE, "mathsweep_emul.cpp")
    if shutil.which("g++") and (force or _newer([src] + hdrs, EMUL_SO)):
        r = subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-DNH_HOSTSIM", "-I" + CSRC,
                            "-I" + os.path.join(ROOT, "include"), src, "-o", EMUL_SO, "-lm", "-lpthread"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed on mathsweep_emul.cpp:\n" + r.stdout)
    out["emul"] = EMUL_SO if os.path.exists(EMUL_SO) else None
    return out


def nchunks(lo, hi):
    return ((hi - lo) + (1 << CHUNK_LOG2) - 1) >> CHUNK_LOG2


def _sums(fn, which, lo, hi, *extra):
    out = np.zeros(nchunks(lo, hi), np.uint64)
    rc = fn(which, ctypes.c_uint32(lo), ctypes.c_uint32(hi), CHUNK_LOG2, out.ctypes.data_as(ctypes.c_void_p), *extra)
    if rc != 0:
        raise RuntimeError("sweep %d over [%#x, %#x) failed: %d" % (which, lo, hi, rc))
    return out


def host_sums(which, lo, hi, threads):
    """Per-chunk checksums of what the reference computes (libm)."""
    return _sums(ctypes.CDLL(HOST_SO).mathsweep_host, which, lo, hi, int(threads))


def emul_sums(which, lo, hi, threads):
    """Per-chunk checksums of agent_math.h's algorithms compiled for the host."""
    return _sums(ctypes.CDLL(EMUL_SO).mathsweep_emul, which, lo, hi, int(threads))


def device_sums(which, lo, hi):
    """Per-chunk checksums (or largest ulp errors x 65536) of the functions on the GPU."""
    return _sums(ctypes.CDLL(DEV_SO).mathsweep_run, which, lo, hi)


def device_raw(which, lo, n):
    out = np.zeros(n, np.uint32)
    rc = ctypes.CDLL(DEV_SO).mathsweep_raw(which, ctypes.c_uint32(lo), ctypes.c_uint32(n), out.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError("raw sweep failed: %d" % rc)
    return out


def host_one(which, b):
    fn = ctypes.CDLL(HOST_SO).mathsweep_host_one
    fn.restype = ctypes.c_uint32
    return fn(which, ctypes.c_uint32(b))


def first_difference(which, lo, hi, got_sums, want_sums, raw):
    """Name the first argument whose result differs, given the chunk checksums of both sides.
    raw(which, lo, n) -> the results of the side under test."""
    bad = np.flatnonzero(got_sums != want_sums)
    if len(bad) == 0:
        return None
    c = int(bad[0])
    b0 = lo + (c << CHUNK_LOG2)
    n = min(1 << CHUNK_LOG2, hi - b0)
    got = raw(which, b0, n)
    for k in range(n):
        w = host_one(which, b0 + k)
        if int(got[k]) != w:
            a = struct.unpack("<f", struct.pack("<I", b0 + k))[0]
            return {"bad_chunks": len(bad), "argument_bits": hex(b0 + k), "argument": a, "got_bits": hex(int(got[k])), "want_bits": hex(w)}
    return {"bad_chunks": len(bad), "note": "checksums differ but no raw result does"}


if __name__ == "__main__":
    print(build(force=True))
