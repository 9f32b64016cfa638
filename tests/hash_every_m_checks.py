"""LowHash0 of a library against the oracle for every window length m = 1 .. 13 through the PRODUCTION window-hash
kernel (its fixed-m instantiations and the generic one), with palindromic flags set on the first and last read.
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
