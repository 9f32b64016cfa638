"""Randomized runs of the two stages against the oracle (seeded: read sets, candidate lists and parameter
draws differ from every other test).  Shared by the emulated pre-flight and the -m gpu tests; the same
loops with other seed ranges are what was run by hand (hundreds of seeds) before the kernels that have
not met a GPU yet were made the default.  Test infrastructure."""
import numpy as np

from shasta_amd import abi
from tests import support

ALIGN_DRAWS = (
    dict(),
    dict(minAlignedMarkerCount=10, minAlignedFraction=0.1, maxSkip=100, maxDrift=100, maxTrim=100),
    dict(maxBand=60, minAlignedMarkerCount=20),
    dict(minEntryCountPerCell=3, deltaY=4, minAlignedMarkerCount=30),
)


def align4(lib, oracle_lib, seeds):
    stored = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        toc, kmer, data7 = support.small_marker_set(n_reads=int(rng.integers(40, 90)), genome_markers=int(rng.integers(3000, 9000)), seed=seed)
        lh = oracle_lib.lowhash0(toc, data7, None, abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30, minFrequency=1))
        cand = lh.candidates[:250]
        o = abi.default_align4_options(**ALIGN_DRAWS[seed % len(ALIGN_DRAWS)])
        x = lib.align4_batch(toc, data7, cand, o, want_ordinals=True)
        y = oracle_lib.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
        support.same_align(x, y)
        stored += len(y.alignment_data)
    return stored


def lowhash0(lib, oracle_lib, seeds):
    compared = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        n_reads = int(rng.integers(20, 200))
        toc, kmer, data7 = support.small_marker_set(n_reads=n_reads, genome_markers=int(rng.integers(2000, 12000)), seed=seed)
        flags = (rng.random(n_reads) < 0.05).astype(np.uint8)
        p = abi.default_lowhash0_params(
            m=int(rng.integers(1, 14)), hashFraction=float(rng.choice([0.005, 0.01, 0.05, 0.2])), minHashIterationCount=int(rng.integers(1, 6)),
            minBucketSize=int(rng.integers(0, 4)), maxBucketSize=int(rng.integers(4, 40)), minFrequency=int(rng.integers(1, 4)),
            log2MinHashBucketCount=int(rng.choice([0, 0, 14])))
        try:
            a = oracle_lib.lowhash0(toc, data7, flags, p)
        except RuntimeError:
            continue                    # the reference's own parameter check (bucket count too small for the marker count)
        b = lib.lowhash0(toc, data7, flags, p)
        support.same_lowhash(a, b)
        compared += 1
    return compared
