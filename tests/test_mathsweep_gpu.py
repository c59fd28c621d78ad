"""ON THE DEVICE, over every float of the domain (tests/tools/mathsweep.hip, built with the product's own arithmetic
flags): the functions whose bit-exactness rests on what gfx950's instructions return -- which the host emulator cannot
see, it substitutes IEEE sqrtf for v_sqrt_f32 / v_rsq_f32:

  exp_f32_magic   == libm's (float)exp((double)a)             all 2.24e9 floats in [-104.5, 89]   (movement.c:1671, :1731)
  sqrt_rn_normal  == correctly rounded sqrtf                   s = 0 and every float in [2^-90, 2^90]   (PFM_Vec2_Len)
  cohesion_t_f32  == (float)(((double)len - 37.5) / 50.0f)    every float in [16, 8192)           (movement.c:1668)
  cohesion_t_f64  == the same                                   every float in [0, 16)
  vlen((a, 0))    == sqrtf(a * a)                              every non-negative float up to 3e38 (the guard + cold IEEE branch)
  v_rsq_f32, v_sqrt_f32: at most 1 ulp from the f64 value over [2^-90, 2^90] -- what the margins of cone_contains_fast /
  cone_test_bf and the one-ulp fix-up of sqrt_rn_normal assume.
The reference side (libm on the host cores) folds into the same per-chunk checksums: tests/tools/mathsweep_host.c."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.tools import mathsweep as ms     # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def built():
    b = ms.build()
    assert b["device"] and b["host"], "tests/tools/_mathsweep.so / _mathsweep_host.so are not built (__graft_entry__.build())"
    return b


def _threads():
    import bench
    return max(1, min(32, bench.usable_cores()))


@pytest.mark.parametrize("name", sorted(ms.SWEEPS))
def test_device_results_equal_libm_for_every_float(built, name):
    which, ranges = ms.SWEEPS[name]
    for lo, hi in ranges:
        got = ms.device_sums(which, lo, hi)
        want = ms.host_sums(which, lo, hi, _threads())
        diff = ms.first_difference(which, lo, hi, got, want, ms.device_raw)
        assert diff is None, (name, hex(lo), hex(hi), diff)


@pytest.mark.parametrize("name", sorted(ms.ULP_SWEEPS))
def test_native_approximations_stay_within_one_ulp(built, name):
    which, ranges = ms.ULP_SWEEPS[name]
    worst = 0.0
    for lo, hi in ranges:
        worst = max(worst, float(ms.device_sums(which, lo, hi).max()) / 65536.0)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "mathsweep_%s.json" % name), "w") as f:
            json.dump({"instruction": name, "max_error_ulp": worst, "range": "[2^-90, 2^90], every float"}, f)
    assert worst <= 1.0, (name, worst)
