"""Live comparison of the restatement with the reference compiled in place
(oracle/_ref).  Skipped when that library has not been built."""
import numpy as np
import pytest

from shasta_amd import abi
from tests import support


def test_reference_codec_selftest(ref_lib):
    ref_lib.test_alignment_compression()


def test_murmur(ref_lib, oracle_lib):
    rng = np.random.default_rng(1)
    for n in list(range(0, 40)) + [64, 100]:
        data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        for seed in (0, 37, 370, 2**40 + 5):
            assert ref_lib.murmur64a(data, seed) == oracle_lib.murmur64a(data, seed)


def test_codec_random(ref_lib, oracle_lib):
    rng = np.random.default_rng(2)
    for trial in range(50):
        n = int(rng.integers(0, 400))
        steps = rng.choice([1, 1, 1, 1, 2, 3, 9, 40, 600, 70000, 3000000], size=(n, 2))
        steps[rng.random(n) < 0.6] = 1
        o = np.cumsum(steps, axis=0).astype(np.uint32)
        a, b = ref_lib.compress(o), oracle_lib.compress(o)
        assert np.array_equal(a, b)
        assert np.array_equal(oracle_lib.decompress(a), o)
        assert np.array_equal(ref_lib.decompress(b), o)


def test_alignment_info_random(ref_lib, oracle_lib):
    rng = np.random.default_rng(3)
    for trial in range(50):
        n = int(rng.integers(1, 300))
        o = np.cumsum(rng.integers(1, 6, size=(n, 2)), axis=0).astype(np.uint32)
        nx, ny = int(o[-1, 0]) + int(rng.integers(1, 50)), int(o[-1, 1]) + int(rng.integers(1, 50))
        a, b = ref_lib.alignment_info(o, nx, ny), oracle_lib.alignment_info(o, nx, ny)
        assert bytes(a)[:49] == bytes(b)[:49]


@pytest.mark.parametrize("seed", [11, 12])
def test_lowhash0_and_align4_on_marker_level_reads(ref_lib, oracle_lib, seed):
    toc, kmer, data7 = support.small_marker_set(n_reads=250, genome_markers=15000, seed=seed)
    flags = np.zeros(250, np.uint8)
    flags[[5, 17]] = 1
    p = abi.default_lowhash0_params(minBucketSize=3, maxBucketSize=30, minFrequency=2)
    a = ref_lib.lowhash0(toc, data7, flags, p, threads=2)
    b = oracle_lib.lowhash0(toc, data7, flags, p)
    support.same_lowhash(a, b)
    assert len(a.candidates) > 100
    cand = a.candidates[:400]
    o = abi.default_align4_options(minAlignedMarkerCount=40)
    x = ref_lib.align4_batch(toc, data7, cand, o)
    y = oracle_lib.align4_batch(toc, data7, cand, o, threads=0)
    keep = (y.status & 0x80) == 0          # component ties: reference order is hash-order dependent
    assert keep.mean() > 0.95
    assert np.array_equal(x.status[keep], (y.status & 0x7f)[keep])
    if keep.all():
        y.status &= 0x7f
        support.same_align(x, y)


def test_kmer_downsampling_hash(ref_lib, oracle_lib):
    # KmerInfo::hash through the reference's Kmer class and MurmurHash2 vs the restatement.
    for k in (4, 7, 10):
        ids = np.arange(1 << (2 * k), dtype=np.uint32)
        assert np.array_equal(ref_lib.kmer_hashes(k), oracle_lib.kmer_hashes(ids, k))


@pytest.mark.parametrize("seed,kw", [
    (21, dict()),
    (22, dict(downsamplingFactor=0.05, minAlignedMarkerCount=40)),
    (23, dict(downsamplingFactor=0.25, bandExtend=2, maxBand=30, minAlignedMarkerCount=20, suppressContainments=1)),
    (24, dict(downsamplingFactor=0.0005, minAlignedMarkerCount=40)),     # mostly empty down-sampled reads
])
def test_align3_on_marker_level_reads(ref_lib, oracle_lib, seed, kw):
    toc, kmer, data7 = support.small_marker_set(n_reads=250, genome_markers=15000, seed=seed)
    p = abi.default_lowhash0_params(minBucketSize=3, maxBucketSize=30, minFrequency=2)
    cand = ref_lib.lowhash0(toc, data7, None, p, threads=2).candidates[:600]
    assert len(cand) > 100
    o = abi.default_align3_options(**kw)
    x = ref_lib.align3_batch(toc, data7, cand, o, threads=4)
    y = oracle_lib.align3_batch(toc, data7, cand, o, threads=0)
    support.same_align(x, y)
    assert np.array_equal(x.compressed_data, y.compressed_data)


@pytest.mark.parametrize("k,seed", [(8, 5), (10, 6), (12, 7)])
def test_marker_finding(ref_lib, oracle_lib, tmp_path, k, seed):
    from shasta_amd import synthetic
    fasta = str(tmp_path / "reads.fasta")
    synthetic.fasta_reads(fasta, n_reads=25, genome_length=40000, mean_length=12000.0, seed=seed)
    z = ref_lib.reads_and_markers_from_fasta(fasta, k=k, probability=0.15, seed=seed)
    toc, data = oracle_lib.find_markers(z["reads_toc"], z["reads_data"], z["base_counts"], k, z["is_marker"])
    assert np.array_equal(toc, z["toc"]) and np.array_equal(data, z["data7"]) and int(toc[-1]) > 10000



@pytest.mark.skipif(not __import__("os").environ.get("SHASTA_SLOW_TESTS"), reason="32 GB of bucket arrays and minutes per run in the reference (SHASTA_SLOW_TESTS=1); its digests are committed, tests/golden/log2_31_digests.json")
@pytest.mark.parametrize("name", ["log2 = 31", "log2 = 40 (capped at 31)"])
def test_reference_at_2_to_the_31_buckets_reproduces_the_committed_digest(ref_lib, name):
    import json
    import os
    from tests.golden import make_log2_31_digest as made
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "log2_31_digests.json")) as f:
        golden = json.load(f)["cases"][name]
    assert made.run(ref_lib, name) == golden


def test_sorted_markers_once_per_run_and_the_alignment_table(ref_lib, oracle_lib):
    # bench.py's CPU leg: computeSortedMarkers once for all reads (as the reference's computeAlignments does) instead of a sort
    # per candidate -- the same alignments -- and computeAlignmentTable in the reference's container against its restatement.
    import numpy as np
    from shasta_amd import abi
    from tests import host_support, support
    toc, kmer, data7 = support.small_marker_set(n_reads=120, genome_markers=9000, seed=88)
    data7 = np.ascontiguousarray(data7, dtype=np.uint8)
    cand = oracle_lib.lowhash0(toc, data7, None, abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30, minFrequency=1)).candidates[:500]
    o = abi.default_align4_options(minAlignedMarkerCount=40)
    a = ref_lib.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=2)
    assert ref_lib.compute_sorted_markers(toc, data7, threads=2) > 0.0
    try:
        b = ref_lib.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=2)
    finally:
        ref_lib.drop_sorted_markers()
    support.same_align(a, b)
    rows = np.array(a.alignment_data, copy=True)
    assert len(rows) > 100
    table_toc, table_values, seconds = ref_lib.alignment_table(rows, 120)
    expected_toc, expected_values = host_support.alignment_table_expected(120, rows)
    assert np.array_equal(table_toc, expected_toc.astype(np.uint64)) and np.array_equal(table_values, expected_values)
