"""The property checker of tests/properties.py against the ORACLE's outputs on a small read set: proves
the checker itself right (it is then applied to the large GPU runs in tests/test_gpu_large_properties.py)."""
import numpy as np
import pytest

from shasta_amd import abi
from tests import properties, support


@pytest.mark.parametrize("seed,lh_kw,al_kw", [
    (101, dict(minBucketSize=2, maxBucketSize=30, minFrequency=2), dict(minAlignedMarkerCount=40)),
    (102, dict(m=5, minBucketSize=0, maxBucketSize=10, minFrequency=1),
     dict(minAlignedMarkerCount=10, minAlignedFraction=0.1, maxSkip=100, maxDrift=100, maxTrim=100, suppressContainments=1)),
])
def test_oracle_outputs_satisfy_the_properties(oracle_lib, seed, lh_kw, al_kw):
    toc, kmer, data7 = support.small_marker_set(n_reads=250, genome_markers=15000, seed=seed)
    flags = np.zeros(250, np.uint8)
    flags[[4, 100]] = 1
    p = abi.default_lowhash0_params(**lh_kw)
    lh = oracle_lib.lowhash0(toc, data7, flags, p)
    assert len(lh.candidates) > 200
    properties.check_lowhash0(toc, flags, p, lh)
    o = abi.default_align4_options(**al_kw)
    al = oracle_lib.align4_batch(toc, data7, lh.candidates, o, want_ordinals=False, threads=0)
    assert len(al.alignment_data) > 20
    properties.check_align4(toc, kmer, lh.candidates, o, al, oracle_lib.decompress)


def test_checker_rejects_a_corrupted_result(oracle_lib):
    toc, kmer, data7 = support.small_marker_set(n_reads=150, genome_markers=9000, seed=103)
    p = abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30)
    lh = oracle_lib.lowhash0(toc, data7, None, p)
    o = abi.default_align4_options(minAlignedMarkerCount=40)
    al = oracle_lib.align4_batch(toc, data7, lh.candidates, o, want_ordinals=False, threads=0)
    al.alignment_data = al.alignment_data.copy()
    al.alignment_data["maxSkip"][0] += 1
    with pytest.raises(AssertionError):
        properties.check_align4(toc, kmer, lh.candidates, o, al, oracle_lib.decompress, sample=len(al.alignment_data))
    lh.candidates = lh.candidates[::-1].copy()
    with pytest.raises(AssertionError):
        properties.check_lowhash0(toc, None, p, lh)
