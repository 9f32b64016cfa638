"""LowHash0 of one library against the oracle for every window length m = 1 .. 13 (the window-hash kernel K1
has fixed-m instantiations, a generic one, and two versions: SHASTA_MI355X_HASH=1 is the one without shared
block transforms).  Run as a script in a process of its own by the tests (the version is fixed per process):

    python tests/hash_versions_check.py <library.so> [reads per set]

Test infrastructure: the oracle is the checker, the library is what is checked."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def sweep(lib, orc, reads=120):
    from shasta_amd import abi
    from tests import support
    checked = 0
    for m in range(1, 14):
        toc, kmer, data7 = support.small_marker_set(n_reads=reads, genome_markers=7000, seed=40 + m)
        flags = np.zeros(reads, np.uint8)
        flags[[1, reads - 1]] = 1
        p = abi.default_lowhash0_params(m=m, hashFraction=0.04, minHashIterationCount=4, minBucketSize=2, maxBucketSize=30, minFrequency=1)
        a = orc.lowhash0(toc, data7, flags, p)
        b = lib.lowhash0(toc, data7, flags, p)
        support.same_lowhash(a, b)
        checked += len(a.candidates)
    return checked


def main():
    from oracle import bindings
    from shasta_amd import lib as libmod
    reads = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    checked = sweep(libmod.Library(sys.argv[1]), bindings.OracleLib(), reads)
    print("candidates compared", checked)
    sys.exit(0 if checked > 8 * reads else 1)


if __name__ == "__main__":
    main()
