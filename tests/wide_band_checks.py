"""Bands of more than 1024 diagonals (Align.maxBand beyond what the banded DP kernels hold; the reference only compares a
component's band with maxBand, src/Align4.cpp:929): the wide DP path of the library against the oracle -- single DP tasks of
1100 .. 9000 diagonals mixed with narrow ones in one call, and Align4 on reads whose alignment drifts by more than 1024 markers
so that its one component needs such a band.  Test infrastructure: the oracle is the checker."""
import numpy as np

from shasta_amd import abi
from tests import adversarial, dp_geometry_checks, support


def dp_tasks(lib, orc, seed=71, widths=(1100, 2500, 40, 9000, 64, 1025, 300), n_range=(600, 1500)):
    rng = np.random.default_rng(seed)
    pieces, spec, at = [], [], 0
    for t, width in enumerate(widths):
        alphabet = (1 << 20) if t % 2 == 0 else 9
        n = int(rng.integers(*n_range))
        genome = rng.integers(0, alphabet, size=2 * n + 400, dtype=np.uint32)
        off = int(rng.integers(0, 200))
        a = dp_geometry_checks.noisy(rng, genome[:n], alphabet)
        b = dp_geometry_checks.noisy(rng, genome[off:off + n], alphabet)
        lo = off - width // 2 + int(rng.integers(-20, 20))
        lo = min(max(lo, -len(b) - width + 1), len(a))
        pieces += [a, b]
        spec.append((at, len(a), at + len(a), len(b), lo, lo + width - 1))
        at += len(a) + len(b)
    kmer = np.concatenate(pieces)
    spec = np.asarray(spec, dtype=np.int64)
    got = lib.banded_dp_many(kmer, spec[:, 0], spec[:, 1], spec[:, 2], spec[:, 3], spec[:, 4], spec[:, 5])
    bad = 0
    for (b0, nx, b1, ny, lo, hi), (y, sy) in zip(spec, got):
        x, sx = orc.banded_dp(kmer[b0:b0 + nx], kmer[b1:b1 + ny], int(lo), int(hi))
        if not (sx == sy and np.array_equal(x, y)):
            bad += 1
            print("MISMATCH wide band: nx %d ny %d band [%d, %d]: score %d / %d, %d / %d markers" % (nx, ny, lo, hi, sx, sy, len(x), len(y)))
    return len(spec), bad


def drifting_reads(seed=72, length=5200, every=4):
    """Read 1 = read 0 with one marker in `every` dropped: the alignment's diagonal moves by one marker every `every`, past 1024
    over the read; plus an ordinary overlapping pair and a read that shares nothing."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, adversarial.A, size=length + 2000, dtype=np.uint32)
    a = genome[:length]
    keep = np.ones(length, bool); keep[np.arange(every, length, every)] = False
    b = a[keep]
    return [a, b, genome[length - 800:length + 1200], rng.integers(0, adversarial.A, size=900, dtype=np.uint32)]


def aligner(lib, orc, max_band=3000, **kw):
    reads = drifting_reads(**kw)
    toc, kmer, data7 = adversarial.build(reads)
    cand = adversarial.all_pairs(len(reads))
    big = 10 ** 6
    o = abi.default_align4_options(maxBand=max_band, maxSkip=big, maxDrift=big, maxTrim=big, minAlignedMarkerCount=10)
    want = orc.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
    got = lib.align4_batch(toc, data7, cand, o, want_ordinals=True)
    support.same_align(want, got)
    # The drifting pair is there, stored, and its alignment spans more than 1024 diagonals.
    rows = want.info_table()
    spans = rows[:, 3 + 8] - rows[:, 3 + 7]              # maxOrdinalOffset - minOrdinalOffset
    assert (spans > 1024).any(), spans
    # With the default maxBand the component is dropped (src/Align4.cpp:929) on both sides alike.
    o2 = abi.default_align4_options(maxSkip=big, maxDrift=big, maxTrim=big, minAlignedMarkerCount=10)
    support.same_align(orc.align4_batch(toc, data7, cand, o2, want_ordinals=True, threads=0), lib.align4_batch(toc, data7, cand, o2, want_ordinals=True))
    return int((spans > 1024).sum())
