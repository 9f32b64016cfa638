"""Sweeps over kernel geometry on the MI355X, against the oracle: the forward DP + traceback over every band class and
matrix shape, one task per call and mixed batches; the production window-hash kernel for every m = 1 .. 13."""
import pytest

pytestmark = pytest.mark.gpu


def test_banded_dp_geometries_against_oracle(gpu_lib, oracle_lib):
    from tests import dp_geometry_checks
    cases, batch = dp_geometry_checks.check(gpu_lib, oracle_lib, seed=11)
    assert cases > 100 and batch > 100


def test_banded_dp_wavefront_with_long_and_short_tasks_of_different_widths(gpu_lib, oracle_lib):
    from tests import dp_geometry_checks
    for seed in (3, 4, 5, 6):
        cases, bad = dp_geometry_checks.straddling_bundles(gpu_lib, oracle_lib, seed, long_tasks=6 + seed, short_tasks=7)
        assert cases == 13 + seed and bad == 0


def test_window_hash_kernel_for_every_m(gpu_lib, oracle_lib):
    from tests import hash_every_m_checks
    assert hash_every_m_checks.sweep(gpu_lib, oracle_lib, reads=120) > 8 * 120
