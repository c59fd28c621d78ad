"""A short run of the randomised GPU-vs-oracle sweep (tests/tools/fuzz_gpu.py): random map shapes,
blocker densities, destinations (island ids of blocked portal tiles included), agent layouts and tick
rates -- fields, integration costs, velocities, positions and status bits all bit-identical."""
import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_cases_are_bit_identical():
    from tests.tools import fuzz_gpu
    assert fuzz_gpu.main(8) == 0
