"""Bands of more than 1024 diagonals on the MI355X (tests/wide_band_checks.py)."""
import pytest

from tests import wide_band_checks

pytestmark = pytest.mark.gpu


def test_dp_tasks_of_more_than_1024_diagonals(gpu_lib, oracle_lib):
    tasks, bad = wide_band_checks.dp_tasks(gpu_lib, oracle_lib, widths=(1100, 2500, 40, 9000, 64, 1025, 300, 20000, 70), n_range=(900, 3000))
    assert tasks == 9 and bad == 0


def test_align4_component_of_more_than_1024_diagonals(gpu_lib, oracle_lib):
    assert wide_band_checks.aligner(gpu_lib, oracle_lib) >= 1
    assert wide_band_checks.aligner(gpu_lib, oracle_lib, max_band=20000, seed=73, length=9000, every=3) >= 1
