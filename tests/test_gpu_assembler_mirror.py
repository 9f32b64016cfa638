"""The two hot functions through shasta_amd.assembler (the reference's Python stage surface) on a
Data/ directory written by the reference's containers; outputs against the oracle.  Named to run
last among the GPU tests."""
import os

import numpy as np
import pytest

import shasta_amd.assembler as shasta
from shasta_amd import abi
from tests import support

@pytest.mark.gpu
def test_hot_functions_through_the_mirror(gpu_lib, ref_lib, oracle_lib, tmp_path, monkeypatch):
    toc, kmer, data7 = support.small_marker_set(n_reads=200, genome_markers=12000, seed=86)
    d = str(tmp_path / "Data")
    os.makedirs(d)
    ref_lib.write_data_dir(d, toc, data7, None)
    monkeypatch.chdir(tmp_path)                                   # the CSV side files go to the run directory
    a = shasta.Assembler()                                        # default prefix "Data/", as in the reference
    a.accessKmers(); a.accessMarkers()
    a.findAlignmentCandidatesLowHash0(m=4, hashFraction=0.01, minHashIterationCount=10, alignmentCandidatesPerRead=20.0,
                                      minBucketSize=2, maxBucketSize=30, minFrequency=2)
    ref = oracle_lib.lowhash0(toc, data7, None, abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30, minFrequency=2))
    stored, _ = ref_lib.open_vector(os.path.join(d, "AlignmentCandidates"), 12)
    assert np.array_equal(stored.view("<u4").reshape(-1, 3)[:, :2], ref.candidate_tuples()[:, :2])
    assert np.array_equal(stored[:, 8], ref.candidate_tuples()[:, 2].astype(np.uint8))
    assert os.path.exists("LowHashBucketHistogram.csv") and os.path.exists("ReadLowHashStatistics.csv")
    a.computeCandidateTable()
    a.accessAlignmentCandidates()
    o = shasta.AlignOptions()
    o.minAlignedMarkerCount = 40
    a.computeAlignments(o, 0)
    al = oracle_lib.align4_batch(toc, data7, ref.candidates, abi.default_align4_options(minAlignedMarkerCount=40), want_ordinals=False, threads=0)
    rows, _ = ref_lib.open_vector(os.path.join(d, "AlignmentData"), 64)
    got = np.frombuffer(rows.tobytes(), dtype=abi.ALIGNMENT_DATA_DTYPE)
    assert len(got) == len(al.alignment_data) > 20
    for field in abi.ALIGNMENT_DATA_DTYPE.names:
        assert np.array_equal(got[field], al.alignment_data[field]), field
    blob, _ = ref_lib.open_vector(os.path.join(d, "CompressedAlignments.data"), 1)
    assert np.array_equal(blob.reshape(-1), al.compressed_data)
