"""-m gpu: pairs of two reads beyond 8 192 markers (the ordinary pair of conf/Nanopore-UL-May2022.conf) through the windowed class
of the cells stage and the sparse path's classes for long reads -- parity with the oracle and with the reference's own Align4, and
the kernel table's word on which kernels the candidates took (tests/long_read_checks.py)."""
import pytest

from tests import long_read_checks

pytestmark = pytest.mark.gpu


def test_pairs_of_two_long_reads_take_the_windowed_class(gpu_lib, oracle_lib, monkeypatch):
    # (the read set is a handful of reads over one small genome: every second pair of markers of the sample is a true match, and the
    # library's estimate of the random background -- right for a real read set -- would send the pairs to the HBM-scratch kernel)
    monkeypatch.setenv("SHASTA_MI355X_MATCH_SHIFT", "20")
    r = long_read_checks.both_long(gpu_lib, oracle_lib)
    assert r["both_long"] >= 60 and r["stored"] >= 15
    assert r["in_the_windowed_class"] == r["both_long"]                    # every such pair starts there
    assert r["in_the_hbm_scratch_kernel"] <= r["both_long"] // 8          # (overlaps of 24 000 markers and more keep more cells than the class's graph holds)


def test_long_reads_against_the_reference(gpu_lib, oracle_lib, ref_lib, monkeypatch):
    monkeypatch.setenv("SHASTA_MI355X_MATCH_SHIFT", "20")
    r = long_read_checks.both_long(gpu_lib, oracle_lib, ref_lib, seed=67)
    assert r["in_the_windowed_class"] == r["both_long"] and r["stored"] >= 15


def test_long_reads_over_a_small_alphabet(gpu_lib, oracle_lib):
    """8 000 distinct ids (the random background of k = 10): the windowed class's cell table overflows or is not tried at all, the
    HBM-scratch kernel and the dense DP answer -- the path of round 5, still equal to the oracle."""
    r = long_read_checks.both_long(gpu_lib, oracle_lib, alphabet_size=8000, lengths=(9000, 12500, 9500, 8300, 4000, 17000), genome_markers=20000)
    assert r["both_long"] >= 15 and r["in_the_hbm_scratch_kernel"] >= 10 and r["stored"] >= 10


def test_full_cell_tables_climb_to_the_reference_answer(gpu_lib, oracle_lib, ref_lib):
    """A few hundred distinct ids: every cell of the alignment matrix collects matches, the windowed class's table fills, then the
    first tables of the HBM-scratch kernel; a full table ends the candidate's counting at once (it walked the whole table for
    every further match before) and the candidate runs again in a larger one."""
    r = long_read_checks.full_tables(gpu_lib, oracle_lib, ref_lib, lengths=(9000, 12000, 8800, 15000), alphabet_size=150)
    assert r["candidates"] == 12 and r["hbm_scratch_launches"] >= 2 and r["hbm_scratch_candidates"] > r["candidates"]
    import os
    if os.environ.get("SHASTA_EMU") != "1":
        assert r["aligner_seconds"] < 30.0, r["aligner_seconds"]          # (twelve candidates; the call's first batch on a new context included)
    r = long_read_checks.full_tables(gpu_lib, oracle_lib, ref_lib)
    assert r["windowed_launches"] >= 1 and r["hbm_scratch_launches"] >= 1


def test_long_reads_with_the_librarys_own_estimate(gpu_lib, oracle_lib):
    r = long_read_checks.both_long(gpu_lib, oracle_lib, seed=68)
    assert r["both_long"] >= 60 and r["stored"] >= 15


@pytest.mark.parametrize("force", ["long", "big"])
def test_every_candidate_through_the_windowed_kernels(gpu_lib, oracle_lib, ref_lib, force):
    assert long_read_checks.forced(gpu_lib, oracle_lib, ref_lib, force) >= 3000
