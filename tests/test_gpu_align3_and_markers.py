"""Align method 3 (Assembler::alignOrientedReads3) and marker finding (MarkerFinder) on the MI355X through the
C ABI: bit-exact against fixtures made by the reference's own code and against the oracle."""
import pytest

from tests import align3_checks, support

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["tiny", "synth"])
@pytest.mark.parametrize("i", [0, 1, 2])
def test_align3_matches_reference_fixture(gpu_lib, name, i):
    align3_checks.golden_fixture(gpu_lib, name, i)


@pytest.mark.parametrize("seed,kw", [
    (21, dict()),
    (22, dict(downsamplingFactor=0.05, minAlignedMarkerCount=40)),
    (23, dict(downsamplingFactor=0.25, bandExtend=2, maxBand=30, minAlignedMarkerCount=20, suppressContainments=1)),
    (24, dict(downsamplingFactor=0.002, minAlignedMarkerCount=40)),
    # Scores other than 6 / -1 / -1 (the reference hands whatever its options hold to SeqAn, src/AssemblerAlign3.cpp:22-33, 120, 257).
    (25, dict(matchScore=3, mismatchScore=-2, gapScore=-3, minAlignedMarkerCount=40)),
    (26, dict(matchScore=10, mismatchScore=-4, gapScore=-1, downsamplingFactor=0.2, minAlignedMarkerCount=40)),
    (27, dict(matchScore=1, mismatchScore=0, gapScore=-2, downsamplingFactor=0.05, minAlignedMarkerCount=20)),
    # Step-2 bands of more than 1024 diagonals (Align.maxBand beyond what the banded DP kernels hold: the wide DP over the band).
    (28, dict(bandExtend=700, maxBand=3000, minAlignedMarkerCount=40)),
    (29, dict(bandExtend=5000, maxBand=20000, downsamplingFactor=0.2, matchScore=3, mismatchScore=-2, gapScore=-3, minAlignedMarkerCount=40)),
])
def test_align3_matches_oracle(gpu_lib, oracle_lib, seed, kw):
    align3_checks.against_oracle(gpu_lib, oracle_lib, seed, kw, n_reads=250, genome_markers=15000, limit=1500)


def test_align3_on_a_context(gpu_lib, oracle_lib):
    align3_checks.context_paths(gpu_lib, oracle_lib)


def test_align3_pairs_with_more_than_1024_and_more_than_8192_downsampled_diagonals(gpu_lib, oracle_lib):
    align3_checks.long_reads(gpu_lib, oracle_lib)


def test_align3_unsupported_options_fail_loudly(gpu_lib):
    align3_checks.rejected_options(gpu_lib)


def test_align3_stage_on_a_data_directory(gpu_lib, oracle_lib, tmp_path, monkeypatch):
    # computeAlignments with alignMethod 3 through the C++ host layer and the Python mirror.
    import shasta_amd.assembler as shasta
    from tests import mirror_checks
    mirror_checks.stages_on_a_data_directory(oracle_lib, tmp_path, monkeypatch, shasta.HOST_SO, 3)


def test_marker_finding(gpu_lib, oracle_lib):
    # SURVEY 8f row 2 (MarkerFinder): also written after the GPU closed; emulation-verified only so far.
    from tests import marker_checks
    marker_checks.golden_fixture(gpu_lib.find_markers)
    for seed, k in ((1, 10), (2, 7), (5, 12)):
        marker_checks.against_oracle(gpu_lib, oracle_lib, seed, k)
    marker_checks.resident_markers_feed_lowhash0(gpu_lib)


def test_find_markers_stage_on_a_data_directory(gpu_lib, tmp_path):
    import shasta_amd.assembler as shasta
    from tests import mirror_checks
    mirror_checks.find_markers_on_a_data_directory(tmp_path, shasta.HOST_SO)
