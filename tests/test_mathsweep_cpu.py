"""The exact-arithmetic building blocks of csrc/agent_math.h whose results must equal the reference's libm calls,
swept over EVERY float of their domain with IEEE host arithmetic (tests/tools/mathsweep_emul.cpp = the header compiled
with -DNH_HOSTSIM) against libm itself (tests/tools/mathsweep_host.c): exp_f32_magic == (float)exp((double)a)
(movement.c:1671, :1731) for all 2.24e9 arguments in [-104.5, 89]; cohesion_t_f32 / _f64 == the double-then-float
expression of movement.c:1668 for every len in [16, 8192) / [0, 16).  The device's own instructions are swept on the
device: tests/test_mathsweep_gpu.py."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.tools import mathsweep as ms     # noqa: E402


@pytest.fixture(scope="module")
def built():
    b = ms.build()
    if not (b["host"] and b["emul"]):
        pytest.skip("no gcc / g++")
    return b


def _threads():
    import bench
    return max(1, min(16, bench.usable_cores()))


def _emul_raw(which, lo, n):
    import ctypes
    import numpy as np
    fn = ctypes.CDLL(ms.EMUL_SO).mathsweep_emul_one
    fn.restype = ctypes.c_uint32
    return np.array([fn(which, ctypes.c_uint32(lo + k)) for k in range(n)], np.uint32)


@pytest.mark.parametrize("name", ["exp", "coh_t_f32", "coh_t_f64"])
def test_every_float_argument_matches_libm(built, name):
    which, ranges = ms.SWEEPS[name]
    swept = 0
    for lo, hi in ranges:
        want = ms.host_sums(which, lo, hi, _threads())
        got = ms.emul_sums(which, lo, hi, _threads())
        assert len(want) == ms.nchunks(lo, hi)
        diff = ms.first_difference(which, lo, hi, got, want, _emul_raw)
        assert diff is None, (name, diff)
        swept += hi - lo
    assert swept >= {"exp": 2_200_000_000, "coh_t_f32": 75_000_000, "coh_t_f64": 1_000_000_000}[name]


def test_the_checksum_sees_a_single_wrong_result(built):
    """One result off by one ulp in one chunk changes that chunk's checksum and first_difference names the argument."""
    import numpy as np
    which, lo = ms.MS_COH_T_F32, ms.bits(100.0)
    hi = lo + (1 << ms.CHUNK_LOG2) + 1000
    want = ms.host_sums(which, lo, hi, 2)
    at = lo + 12345

    def raw(w, b0, n):
        r = _emul_raw(w, b0, n)
        if b0 <= at < b0 + n:
            r[at - b0] += 1
        return r
    got = want.copy()
    got[0] += np.uint64(2 * at + 1)          # what one extra ulp adds to the fold
    d = ms.first_difference(which, lo, hi, got, want, raw)
    assert d and d["argument_bits"] == hex(at) and d["bad_chunks"] == 1
