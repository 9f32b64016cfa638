"""Inputs the synthetic read sets never produce -- empty and one-marker reads, tandem repeats that
overflow every table, alphabets of six k-mers, identical and contained reads, reads of 20 000 markers,
degenerate LowHash0 parameters -- through the library (GPU or emulated build) and the oracle.
Found so far: the member list of the cells stage was sized for one retry per candidate (tandem
repeats climb two classes)."""
import numpy as np

from shasta_amd import abi, synthetic
from tests import support

A = 1 << 20


def build(reads):
    """reads: strand-0 kmer-id arrays; strand 1 = reversed with the low bit flipped (an involution)."""
    toc, ids = [0], []
    for r in reads:
        r = np.asarray(r, np.uint32)
        ids += [r, (r[::-1] ^ np.uint32(1)).astype(np.uint32)]
        toc += [toc[-1] + len(r), toc[-1] + 2 * len(r)]
    toc = np.array(toc, np.uint64)
    kmer = np.concatenate(ids) if ids else np.zeros(0, np.uint32)
    return toc, kmer, synthetic.pack_markers(toc, kmer)


def all_pairs(n):
    r0, r1, s = [], [], []
    for i in range(n):
        for j in range(i + 1, n):
            for strand in (0, 1):
                r0.append(i); r1.append(j); s.append(strand)
    return abi.make_pairs(r0, r1, s)


def read_sets(long_reads=True):
    rng = np.random.default_rng(7)
    genome = rng.integers(0, A, size=6000, dtype=np.uint32)
    yield "degenerate lengths", [genome[:0], genome[:1], genome[:3], genome[:50], genome[10:900], genome[:3000],
                                 genome[100:5100], genome[2000:2001]]
    unit = rng.integers(0, A, size=7, dtype=np.uint32)
    rep = np.tile(unit, 400)
    yield "tandem repeats", [rep[:1500], rep[3:2000], np.concatenate([genome[:500], rep[:800], genome[500:900]]),
                             np.concatenate([genome[100:700], rep[:600]]), np.full(900, 12345, np.uint32),
                             np.full(1200, 12345, np.uint32)]
    small = rng.integers(0, 6, size=4000, dtype=np.uint32)
    yield "alphabet of six", [small[:800], small[100:1100], small[300:1500], small[::2][:700]]
    yield "identical, contained, reversed", [genome[:1200], genome[:1200].copy(), genome[200:800],
                                             genome[:1200][::-1] ^ np.uint32(1)]
    if long_reads:
        big = rng.integers(0, A, size=30000, dtype=np.uint32)
        noisy = lambda x: x[rng.random(len(x)) < 0.8]
        yield "long reads", [noisy(big[:9000]), noisy(big[4000:13500]), noisy(big[8000:8400]), noisy(big[:20000]),
                             noisy(big[15000:15100])]


def aligners(lib, oracle_lib, long_reads=True):
    for name, reads in read_sets(long_reads):
        toc, kmer, data7 = build(reads)
        cand = all_pairs(len(reads))
        o4 = abi.default_align4_options(minAlignedMarkerCount=10)
        x = oracle_lib.align4_batch(toc, data7, cand, o4, want_ordinals=True, threads=0)
        y = lib.align4_batch(toc, data7, cand, o4, want_ordinals=True)
        ties = (x.status & 0x80) != 0
        assert np.array_equal(x.status & 0x80, y.status & 0x80), name
        assert np.array_equal(x.status[~ties], y.status[~ties]), name
        if not ties.any():
            support.same_align(x, y)
        o3 = abi.default_align3_options(minAlignedMarkerCount=10)
        a = oracle_lib.align3_batch(toc, data7, cand, o3, want_ordinals=True, threads=0)
        b = lib.align3_batch(toc, data7, cand, o3, want_ordinals=True)
        support.same_align(a, b)
        assert np.array_equal(a.compressed_data, b.compressed_data), name


def lowhash0(lib, oracle_lib):
    toc, kmer, data7 = support.small_marker_set(n_reads=120, genome_markers=8000, seed=3)
    rng = np.random.default_rng(11)
    P = abi.default_lowhash0_params
    cases = [(None, P(m=m, minBucketSize=2, maxBucketSize=30)) for m in (1, 2, 7)]
    cases += [
        (None, P(hashFraction=1.0, minHashIterationCount=1, minBucketSize=2, maxBucketSize=1000)),
        (None, P(log2MinHashBucketCount=20, minBucketSize=2, maxBucketSize=30)),
        (None, P(minBucketSize=0, maxBucketSize=2, minFrequency=1)),
        (None, P(minBucketSize=2, maxBucketSize=30, minFrequency=9)),
        (None, P(minHashIterationCount=0, alignmentCandidatesPerRead=3.0, minBucketSize=2, maxBucketSize=30)),
        (np.ones(120, np.uint8), P(minBucketSize=2, maxBucketSize=30)),
        ((rng.random(120) < 0.5).astype(np.uint8), P(minBucketSize=2, maxBucketSize=30)),
    ]
    for flags, p in cases:
        support.same_lowhash(lib.lowhash0(toc, data7, flags, p), oracle_lib.lowhash0(toc, data7, flags, p))
    # A forced bucket count below the minimum is an error on both sides (src/LowHash0.cpp:85-96).
    import pytest
    with pytest.raises(RuntimeError):
        lib.lowhash0(toc, data7, None, P(log2MinHashBucketCount=6))
    # Empty reads, reads shorter than m, a single read, forty copies of one read.
    g = rng.integers(0, A, size=3000, dtype=np.uint32)
    short = [g[s:s + n] for s, n in zip(rng.integers(0, 2000, size=12), [0, 1, 2, 3, 4, 5, 300, 500, 0, 700, 3, 900])]
    for reads, p in ((short, P(minBucketSize=1, maxBucketSize=30, minFrequency=1)),
                     ([g[:800]], P(minBucketSize=1, maxBucketSize=30, minFrequency=1)),
                     ([g[:600]] * 40, P(minBucketSize=2, maxBucketSize=100, minFrequency=2))):
        t, _, d = build(reads)
        support.same_lowhash(lib.lowhash0(t, d, None, p), oracle_lib.lowhash0(t, d, None, p))


def task_list_overflow(lib, oracle_lib, monkeypatch):
    """More DP tasks than the list was sized for (in production: thousands of small components per
    batch, e.g. minEntryCountPerCell = 1 on repeat-rich reads): the cells stage runs again with the
    exact count.  Forced here by a tiny first guess."""
    toc, kmer, data7 = support.small_marker_set(n_reads=100, genome_markers=8000, seed=9)
    p = abi.default_lowhash0_params(minBucketSize=3, maxBucketSize=30, minFrequency=2)
    cand = oracle_lib.lowhash0(toc, data7, None, p).candidates[:300]
    o = abi.default_align4_options(minAlignedMarkerCount=40)
    ref = oracle_lib.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
    monkeypatch.setenv("SHASTA_MI355X_INITIAL_TASKS", "7")
    out = lib.align4_batch(toc, data7, cand, o, want_ordinals=True)
    monkeypatch.delenv("SHASTA_MI355X_INITIAL_TASKS")
    assert (ref.status == abi.SHASTA_ALIGN_STORED).sum() > 100
    if not (ref.status & 0x80).any():
        support.same_align(ref, out)
    assert out.dp_cell_count == ref.dp_cell_count

