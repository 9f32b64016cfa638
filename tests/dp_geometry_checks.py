"""Banded DP of a library against the oracle over a sweep of band geometries (every band class,
tiny and ragged matrices, bands hanging over every corner, small alphabets = score ties): one task at a
time (`sweep`) and many tasks of mixed geometry in one batch (`many`: several tasks to a wavefront, as in
an Align4 batch).  Test infrastructure: the oracle is the checker, the library is what is checked."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTHS = (1, 2, 3, 16, 20, 31, 32, 33, 40, 48, 49, 64, 65, 70, 80, 81, 100, 128, 129, 256, 300, 512, 513, 1000)


def noisy(rng, x, alphabet):
    keep = rng.random(len(x)) > 0.06
    y = x[keep].copy()
    sub = rng.random(len(y)) < 0.08
    y[sub] = rng.integers(0, alphabet, size=int(sub.sum()), dtype=np.uint32)
    return y


def sweep(lib, orc, seed, trials=6):
    rng = np.random.default_rng(seed)
    cases = bad = 0
    for width in WIDTHS:
        for trial in range(trials):
            alphabet = (1 << 20) if trial % 2 == 0 else 12
            n = int(rng.integers(5, 900)) if trial < 4 else int(rng.integers(1, 40))
            m = max(1, n + int(rng.integers(-(n // 2), n // 2 + 1)))
            genome = rng.integers(0, alphabet, size=n + m + 1400, dtype=np.uint32)
            off = int(rng.integers(0, 300))
            a = noisy(rng, genome[:n], alphabet)
            b = noisy(rng, genome[off:off + m], alphabet)
            if len(a) == 0 or len(b) == 0:
                continue
            center = off if trial % 3 else -off
            lo = center + int(rng.integers(-30, 30)) - width // 2
            x, sx = orc.banded_dp(a, b, lo, lo + width - 1)
            y, sy = lib.banded_dp(a, b, lo, lo + width - 1)
            cases += 1
            if not (sx == sy and np.array_equal(x, y)):
                bad += 1
                print("MISMATCH width %d trial %d nx %d ny %d bandMin %d: score %d / %d, %d / %d markers"
                      % (width, trial, len(a), len(b), lo, sx, sy, len(x), len(y)))
    return cases, bad


def many(lib, orc, seed, tasks=140):
    """Tasks of every band class in ONE call, so that wavefronts hold several tasks of different geometry (the path of an
    Align4 batch: sorted by class and length, 4 / 2 / 1 tasks per wavefront); every task against the oracle."""
    rng = np.random.default_rng(seed)
    pieces, spec = [], []
    at = 0
    for t in range(tasks):
        width = int(rng.choice([1, 5, 20, 32, 33, 40, 48, 49, 50, 64, 65, 70, 80, 81, 100, 128, 200, 256, 300, 512, 600, 1000], p=[.04, .04, .1, .04, .04, .1, .04, .04, .1, .04, .04, .08, .04, .04, .06, .03, .04, .02, .02, .02, .02, .01]))
        alphabet = (1 << 20) if t % 2 == 0 else 9
        n = int(rng.integers(3, 700)) if t % 5 else int(rng.integers(700, 1500))
        m = max(1, n + int(rng.integers(-(n // 2), n // 2 + 1)))
        genome = rng.integers(0, alphabet, size=n + m + 400, dtype=np.uint32)
        off = int(rng.integers(0, 200))
        a = noisy(rng, genome[:n], alphabet)
        b = noisy(rng, genome[off:off + m], alphabet)
        if len(a) == 0 or len(b) == 0:
            continue
        center = off if t % 3 else -off
        lo = center + int(rng.integers(-20, 20)) - width // 2
        lo = max(lo, -len(b) - width + 1)
        lo = min(lo, len(a))
        hi = lo + width - 1
        if hi < -len(b) or lo > len(a):
            continue
        pieces += [a, b]
        spec.append((at, len(a), at + len(a), len(b), lo, hi))
        at += len(a) + len(b)
    kmer = np.concatenate(pieces)
    spec = np.asarray(spec, dtype=np.int64)
    got = lib.banded_dp_many(kmer, spec[:, 0], spec[:, 1], spec[:, 2], spec[:, 3], spec[:, 4], spec[:, 5])
    bad = 0
    for (b0, nx, b1, ny, lo, hi), (y, sy) in zip(spec, got):
        x, sx = orc.banded_dp(kmer[b0:b0 + nx], kmer[b1:b1 + ny], int(lo), int(hi))
        if not (sx == sy and np.array_equal(x, y)):
            bad += 1
            print("MISMATCH in a batch: nx %d ny %d band [%d, %d]: score %d / %d, %d / %d markers" % (nx, ny, lo, hi, sx, sy, len(x), len(y)))
    return len(spec), bad


def straddling_bundles(lib, orc, seed, long_tasks=6, short_tasks=7):
    """One band class (widths 49 .. 64: 16 lanes x 4 diagonals, four tasks to a wavefront) with long tasks whose lanes are all
    whole (width 64, 60) and short tasks with a partly filled lane (widths 50, 54), the counts chosen so that one wavefront
    holds tasks of both kinds -- its trace is sized by the longest of its tasks, and its steady loop covers only the
    iterations that are steady for every task."""
    rng = np.random.default_rng(seed)
    pieces, spec = [], []
    at = 0
    for t in range(long_tasks + short_tasks):
        long_task = t < long_tasks
        width = int(rng.choice([64, 60])) if long_task else int(rng.choice([50, 54]))
        n = int(rng.integers(900, 1400)) if long_task else int(rng.integers(120, 300))
        alphabet = (1 << 20) if t % 2 == 0 else 9
        genome = rng.integers(0, alphabet, size=2 * n + 200, dtype=np.uint32)
        off = int(rng.integers(0, 60))
        a = noisy(rng, genome[:n], alphabet)
        b = noisy(rng, genome[off:off + n], alphabet)
        lo = off - width // 2 + int(rng.integers(-8, 8))
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
            print("MISMATCH in a straddling bundle: nx %d ny %d band [%d, %d]: score %d / %d" % (nx, ny, lo, hi, sx, sy))
    return len(spec), bad


def check(lib, orc, seed=11, tasks=140, trials=6):
    cases, bad = sweep(lib, orc, seed, trials)
    assert bad == 0 and cases >= 12 * trials, (cases, bad)
    batch, bad_in_batch = many(lib, orc, seed + 100, tasks)
    assert bad_in_batch == 0 and batch >= tasks * 2 // 3, (batch, bad_in_batch)
    return cases, batch
