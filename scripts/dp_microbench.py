#!/usr/bin/env python3
"""Times the banded DP kernels (K10: forward kernel of every band class + traceback) on synthetic tasks through
the unit seam shasta_mi355x_banded_dp_many, without the rest of the Align4 stage: the quick A/B of a kernel edit.

    python scripts/dp_microbench.py [--tasks 20000] [--length 1500] [--repeat 3]
    SHASTA_MI355X_LIBRARY=<another build> python scripts/dp_microbench.py ...      # the A side of an A/B

Each task aligns two noisy copies of one random marker sequence inside a band around their true diagonal; the band
widths cover the band classes in the proportions of a real batch (mostly <= 64).  Prints one JSON line per repeat:
GCUPS (DP cells = nx x band width per second) per class from HIP events around each launch."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def tasks_of(n, length, seed):
    rng = np.random.default_rng(seed)
    widths = rng.choice([24, 50, 100, 200, 400, 800], size=n, p=[.35, .45, .12, .05, .02, .01])
    pieces, spec, at = [], [], 0
    for t in range(n):
        m = int(max(200, rng.normal(length, length / 4)))
        shift = int(rng.integers(0, m // 2))
        base = rng.integers(0, 1 << 16, size=m + shift, dtype=np.uint32)
        a = base[:m][rng.random(m) > 0.03]
        b = base[shift:][rng.random(m) > 0.03]
        w = int(widths[t])
        lo = shift - w // 2                              # a[i] and b[j] come from base[i] and base[shift + j]: diagonal i - j = shift
        lo = min(max(lo, -len(b) - w + 1), len(a))
        pieces += [a, b]
        spec.append((at, len(a), at + len(a), len(b), lo, lo + w - 1))
        at += len(a) + len(b)
    return np.concatenate(pieces), np.asarray(spec, dtype=np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", type=int, default=20000)
    ap.add_argument("--length", type=int, default=1500)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--library", default=None, help="default: the product library")
    args = ap.parse_args()
    import shasta_amd
    from shasta_amd import lib as libmod
    lib = libmod.Library(args.library) if args.library else shasta_amd.load()
    kmer, spec = tasks_of(args.tasks, args.length, 1)
    names = ["<=32", "<=48", "<=64", "<=80", "<=128", "<=256", "<=512", "<=1024"]
    for r in range(args.repeat + 1):
        counts, scores, seconds, cells = lib.banded_dp_many(kmer, spec[:, 0], spec[:, 1], spec[:, 2], spec[:, 3], spec[:, 4], spec[:, 5], timing=True)
        if r == 0:
            continue                                     # warm-up (allocations, the start-up comparison of the two versions)
        line = {"tasks": args.tasks, "aligned_markers": int(counts.sum()),
                "forward_ms": {names[c]: round(1e3 * seconds[c], 3) for c in range(8) if cells[c]},
                "forward_gcups": {names[c]: round(float(cells[c]) / seconds[c] / 1e9, 1) for c in range(8) if cells[c] and seconds[c] > 0},
                "forward_total_ms": round(1e3 * float(seconds[:8].sum()), 3), "traceback_ms": round(1e3 * float(seconds[8]), 3)}
        print(json.dumps(line))


if __name__ == "__main__":
    main()
