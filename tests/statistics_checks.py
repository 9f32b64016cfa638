"""LowHash0's per-read statistics (src/LowHash0.cpp:386-393) through both forms of K3 (shasta_amd/csrc/lowhash0.hip): keys, a
partition pass and LDS counts (readStatisticsKernel: the default), and an atomic per record (SHASTA_MI355X_STATISTICS_ATOMICS=1) --
on read sets with enough reads for several partitions of the table and enough records for several spans of keys.  Shared by the
-m gpu tests and their pre-flight on the emulated build."""
import os

from shasta_amd import abi
from tests import support


def several_partitions_and_spans(lib, orc, n_reads=1400, cases=((0.25, 3), (0.05, 6))):
    toc, kmer, data7 = support.small_marker_set(n_reads=n_reads, genome_markers=60000, seed=4)
    rows = 0
    for fraction, iterations in cases:
        p = abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=12, minFrequency=1)
        p.hashFraction = fraction
        p.minHashIterationCount = iterations
        want = orc.lowhash0(toc, data7, None, p, threads=0)
        support.same_lowhash(lib.lowhash0(toc, data7, None, p), want)
        os.environ["SHASTA_MI355X_STATISTICS_ATOMICS"] = "1"
        try:
            support.same_lowhash(lib.lowhash0(toc, data7, None, p), want)
        finally:
            del os.environ["SHASTA_MI355X_STATISTICS_ATOMICS"]
        rows += int((want.statistics.reshape(-1, 3).sum(axis=1) > 0).sum())
    return rows
