"""The DP tie policy as a switch (shasta_amd/csrc/align4_dp.hpp, DpTie): the library under each of its two compiled alternative
policies (SHASTA_MI355X_DP_TIE_POLICY=2: diagonal >= horizontal >= vertical, first maximum -- the nearest other reading of SeqAn's
_maxScore(left, right) -- and 3: the same with the last maximum) against the oracle set to the same policy
(oracle/banded_dp.hpp, tiePolicyByIndex) -- on tie-heavy DP tasks of every band class and on the whole aligner -- and the
proof that the two policies really differ on those inputs.  Test infrastructure: the oracle is the checker."""
import os

import numpy as np

from shasta_amd import abi
from tests import dp_geometry_checks, support

ALTERNATIVES = (2, 3)


class policy:
    """Library and oracle both on tie policy `index` inside the block, both back on the default after it."""
    def __init__(self, orc, index):
        self.orc, self.index = orc, index

    def __enter__(self):
        self.previous = os.environ.get("SHASTA_MI355X_DP_TIE_POLICY")
        os.environ["SHASTA_MI355X_DP_TIE_POLICY"] = str(self.index)
        self.orc.set_tie_policy(self.index)

    def __exit__(self, *exc):
        self.orc.set_tie_policy(0)
        if self.previous is None:
            del os.environ["SHASTA_MI355X_DP_TIE_POLICY"]
        else:
            os.environ["SHASTA_MI355X_DP_TIE_POLICY"] = self.previous


def tie_heavy_tasks(seed, tasks=60):
    """Small-alphabet sequences (score ties in almost every cell) and exact repeats (end cells that tie), every band class."""
    rng = np.random.default_rng(seed)
    pieces, spec, at = [], [], 0
    for t in range(tasks):
        width = int(rng.choice([20, 32, 40, 48, 60, 64, 80, 100, 128, 200, 300, 600, 1000]))
        alphabet = int(rng.choice([3, 5, 9]))
        n = int(rng.integers(30, 500))
        if t % 4 == 0:
            unit = rng.integers(0, alphabet, size=int(rng.integers(2, 7)), dtype=np.uint32)      # a tandem repeat: many equal-score end cells
            a = np.tile(unit, n // len(unit) + 1)[:n]
            b = a[: max(1, n - int(rng.integers(0, 20)))].copy()
            off = 0
        else:
            genome = rng.integers(0, alphabet, size=2 * n + 300, dtype=np.uint32)
            off = int(rng.integers(0, 100))
            a = dp_geometry_checks.noisy(rng, genome[:n], alphabet)
            b = dp_geometry_checks.noisy(rng, genome[off:off + n], alphabet)
        if len(a) == 0 or len(b) == 0:
            continue
        lo = off - width // 2 + int(rng.integers(-5, 5))
        lo = min(max(lo, -len(b) - width + 1), len(a))
        pieces += [a, b]
        spec.append((at, len(a), at + len(a), len(b), lo, lo + width - 1))
        at += len(a) + len(b)
    return np.concatenate(pieces), np.asarray(spec, dtype=np.int64)


def dp_tasks_under_the_alternative_policy(lib, orc, seed=21, alternative=2):
    kmer, spec = tie_heavy_tasks(seed)

    def oracle_results():
        return [orc.banded_dp(kmer[b0:b0 + nx], kmer[b1:b1 + ny], int(lo), int(hi)) for b0, nx, b1, ny, lo, hi in spec]

    base = oracle_results()
    got0 = lib.banded_dp_many(kmer, spec[:, 0], spec[:, 1], spec[:, 2], spec[:, 3], spec[:, 4], spec[:, 5])
    with policy(orc, alternative):
        want = oracle_results()
        got = lib.banded_dp_many(kmer, spec[:, 0], spec[:, 1], spec[:, 2], spec[:, 3], spec[:, 4], spec[:, 5])
    bad = sum(1 for (x, sx), (y, sy) in zip(want, got) if not (sx == sy and np.array_equal(x, y)))
    bad0 = sum(1 for (x, sx), (y, sy) in zip(base, got0) if not (sx == sy and np.array_equal(x, y)))
    differ = sum(1 for (x, sx), (y, sy) in zip(base, want) if not np.array_equal(x, y))
    assert all(sx == sy for (x, sx), (y, sy) in zip(base, want))        # the optimal score does not depend on the policy
    return len(spec), bad0, bad, differ


def aligner_under_the_alternative_policy(lib, orc, reads=120, candidates=300, alternative=2):
    toc, kmer, data7 = support.small_marker_set(n_reads=reads, genome_markers=9000, seed=31)
    p = abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30, minFrequency=1)
    cand = orc.lowhash0(toc, data7, None, p).candidates[:candidates]
    o = abi.default_align4_options(minAlignedMarkerCount=40)
    base = orc.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
    with policy(orc, alternative):
        want = orc.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
        got = lib.align4_batch(toc, data7, cand, o, want_ordinals=True)
    support.same_align(want, got)
    differ = sum(1 for x, y in zip(base.per_candidate(), want.per_candidate()) if x != y)
    return len(cand), differ


def unknown_policy_is_refused(lib):
    kmer, spec = tie_heavy_tasks(5, tasks=4)
    previous = os.environ.get("SHASTA_MI355X_DP_TIE_POLICY")
    os.environ["SHASTA_MI355X_DP_TIE_POLICY"] = "7"
    try:
        lib.banded_dp_many(kmer, spec[:, 0], spec[:, 1], spec[:, 2], spec[:, 3], spec[:, 4], spec[:, 5])
    except RuntimeError as e:
        return "tie polic" in str(e)
    finally:
        if previous is None:
            del os.environ["SHASTA_MI355X_DP_TIE_POLICY"]
        else:
            os.environ["SHASTA_MI355X_DP_TIE_POLICY"] = previous
    return False
