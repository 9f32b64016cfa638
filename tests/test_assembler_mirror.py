"""shasta_amd.assembler: the reference's Python stage surface (names and arguments of
src/PythonModule.cpp) over a Data/ directory.  CPU part: option names, the table step, error
behaviour (the GPU part is tests/test_gpu_zz_assembler_mirror.py)."""
import os
import re

import numpy as np
import pytest

import shasta_amd.assembler as shasta
from shasta_amd import abi
from tests import host_support, support

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_align_options_has_the_reference_attribute_names():
    # Attribute names as exposed by the reference's pybind11 class (src/PythonModule.cpp:87-106).
    names = ["alignMethod", "maxSkip", "maxDrift", "maxTrim", "maxMarkerFrequency", "minAlignedMarkerCount",
             "minAlignedFraction", "matchScore", "mismatchScore", "gapScore", "downsamplingFactor", "bandExtend",
             "maxBand", "sameChannelReadAlignmentSuppressDeltaThreshold", "suppressContainments", "align4DeltaX",
             "align4DeltaY", "align4MinEntryCountPerCell", "align4MaxDistanceFromBoundary"]
    o = shasta.AlignOptions()
    for n in names:
        assert hasattr(o, n), n
    assert (o.align4DeltaX, o.align4DeltaY, o.align4MinEntryCountPerCell, o.align4MaxDistanceFromBoundary) == (200, 10, 10, 100)


@pytest.mark.parametrize("build", ["emulated", pytest.param("mi355x", marks=pytest.mark.gpu)])
def test_candidate_table_and_errors_through_the_mirror(ref_lib, oracle_lib, tmp_path, build):
    # (The candidate table is built on the device: the emulated build here, the product on the GPU box.)
    toc, kmer, data7 = support.small_marker_set(n_reads=100, genome_markers=7000, seed=85)
    d = str(tmp_path / "Data")
    os.makedirs(d)
    if build == "emulated":
        host_support.emulated_build()
    a = shasta.Assembler(d + "/", hostLibrary=host_support.EMU_HOST_SO if build == "emulated" else shasta.HOST_SO)
    with pytest.raises(RuntimeError, match="Error accessing"):
        a.accessMarkers()
    ref_lib.write_data_dir(d, toc, data7, None)
    a.accessKmers()
    a.accessMarkers()
    with pytest.raises(RuntimeError, match="Error accessing"):
        a.accessAlignmentCandidates()
    cand = oracle_lib.lowhash0(toc, data7, None, abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30)).candidates
    host_support.HostShim(host_support.EMU_SHIM if build == "emulated" else host_support.SHIM).store_candidates(d, cand)
    a.accessAlignmentCandidates()
    a.computeCandidateTable()
    t, _ = ref_lib.open_vector(os.path.join(d, "CandidateTable.toc"), 8)
    dta, _ = ref_lib.open_vector(os.path.join(d, "CandidateTable.data"), 8)
    toc_expected, data_expected = host_support.alignment_table_expected(100, cand, np.uint64)
    assert np.array_equal(t.view("<u8").reshape(-1), toc_expected) and np.array_equal(dta.view("<u8").reshape(-1), data_expected)
    o = shasta.AlignOptions()
    o.alignMethod = 1
    with pytest.raises(RuntimeError, match="alignMethod 3 and 4 only"):
        a.computeAlignments(o, 0)
    o.alignMethod = 3                                   # needs k: the size of Data/Kmers
    with pytest.raises(RuntimeError, match="Kmers"):
        a.computeAlignments(o, 0)
