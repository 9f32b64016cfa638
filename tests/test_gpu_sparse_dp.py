"""The sparse form of the banded alignment on the MI355X (tests/sparse_checks.py)."""
import pytest

from tests import sparse_checks

pytestmark = pytest.mark.gpu


def test_dp_tasks_with_the_sparse_path_on_and_off_and_under_every_compiled_tie_policy(gpu_lib, oracle_lib):
    tasks, clean_share, tie_heavy_share = sparse_checks.dp_tasks(gpu_lib, oracle_lib)
    assert tasks >= 50
    assert clean_share > 0.6               # most cells of clean tasks never reach the dense kernels ...
    assert tie_heavy_share < 0.3           # ... and tie-heavy ones are left to them


def test_aligner_with_the_sparse_path_on_and_off(gpu_lib, oracle_lib):
    assert sparse_checks.aligner(gpu_lib, oracle_lib) > 0.6


def test_locally_ambiguous_tasks_through_the_anchor_kernel(gpu_lib, oracle_lib):
    # align4_anchor.hpp: tasks whose optimal chains differ only around doubled markers and short tandem copies -- the sparse path alone
    # leaves nearly all of them to the dense kernels, the anchor kernel solves the rectangles between the anchors and leaves a fraction.
    runs, cells_all, cells_sparse, cells_anchored = sparse_checks.anchored_tasks(gpu_lib, oracle_lib, seeds=(3, 4, 5, 6), tasks=40)
    assert runs >= 400 and cells_sparse > 0.8 * cells_all and cells_anchored < 0.4 * cells_all


def test_tiny_tasks_through_the_sparse_and_anchor_kernels(gpu_lib, oracle_lib):
    assert sparse_checks.tiny_tasks(gpu_lib, oracle_lib) >= 1200


def test_the_wave_kernel_against_the_forms_it_can_be_switched_to(gpu_lib, oracle_lib):
    # align4_chainwave.hpp: tasks of every capacity class; the lane-per-task kernel, the wave kernel with its own ordering of the hits,
    # the larger classes on the side stream.
    assert sparse_checks.wave_kernel_forms(gpu_lib, oracle_lib) >= 200


def test_the_anchor_kernel_second_launch(gpu_lib, oracle_lib):
    # Rectangles of 441 to 90 601 cells and one with sides of 3 201 markers between two anchors: with the second launch only the last
    # one's task is left to the dense kernels, without it the last five.
    with_second, without = sparse_checks.anchor_kernel_second_launch(gpu_lib, oracle_lib)
    assert 0 < with_second < without


def test_a_call_without_ordinals(gpu_lib, oracle_lib):
    assert sparse_checks.without_ordinals(gpu_lib, oracle_lib) >= 300


def test_long_dense_paths(gpu_lib, oracle_lib):
    assert sparse_checks.long_dense_paths(gpu_lib, oracle_lib) == 20
