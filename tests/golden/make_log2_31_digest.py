#!/usr/bin/env python3
"""Digests of the REFERENCE's own LowHash0 (oracle/_ref: src/LowHash0.cpp compiled in place) at log2MinHashBucketCount = 31
and at a request of 40 (capped at 31 with a message, src/LowHash0.cpp:73-98) -- the values of the human-genome runs, where bit 31
of the hash takes part in neither the bucket id nor the match key (src/LowHash0.hpp:99-105).  The reference allocates two
arrays of 2^31 eight-byte entries (32 GB) and sweeps them every iteration: minutes per run, not a CI test.  This script is
that run; tests/test_oracle_golden.py checks the restatement (which the GPU tests of these values compare with) against the
digests it wrote, and `SHASTA_SLOW_TESTS=1 pytest tests/test_oracle_vs_ref.py -k log2_31` repeats the reference run.

    python tests/golden/make_log2_31_digest.py        # needs /root/reference (oracle/_ref built) and about 40 GB of RAM
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CASES = ("log2 = 31", "log2 = 40 (capped at 31)")
OUT = os.path.join(ROOT, "tests", "golden", "log2_31_digests.json")


def digest(result):
    h = hashlib.sha256()
    for a in (result.candidate_tuples(), result.statistics, result.high_frequency, result.histogram):
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes())
    return {"sha256": h.hexdigest(), "candidates": int(len(result.candidates)), "high_frequency": [int(x) for x in result.high_frequency],
            "statistics_sum": [int(x) for x in np.asarray(result.statistics).sum(axis=0)]}


def run(lib, name):
    from tests import adversarial, support
    toc, kmer, data7 = support.small_marker_set(n_reads=120, genome_markers=8000, seed=3)
    flags, p = adversarial.lowhash_cases()[name]
    return digest(lib.lowhash0(toc, data7, flags, p))


if __name__ == "__main__":
    from oracle import bindings
    ref = bindings.RefLib()
    out = {"made_by": "tests/golden/make_log2_31_digest.py: oracle/_ref (the reference's LowHash0.cpp compiled in place) on support.small_marker_set(120, 8000, seed=3)",
           "cases": {name: run(ref, name) for name in CASES}}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
