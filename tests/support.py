"""Shared helpers for the tests: golden fixtures and small seeded inputs."""
import os

import numpy as np

from shasta_amd import abi, synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Must match tests/golden/make_golden.py.
LOWHASH_PARAM_SETS = [
    dict(),
    dict(m=5, minBucketSize=2, maxBucketSize=5, minFrequency=3),
    dict(m=3, hashFraction=0.05, minHashIterationCount=0, alignmentCandidatesPerRead=12.0),
]
ALIGN_OPTION_SETS = [
    dict(),
    dict(minAlignedMarkerCount=10, minAlignedFraction=0.1, maxSkip=100, maxDrift=100, maxTrim=100,
         suppressContainments=1),
]

# Must match tests/golden/make_golden_align3.py (align method 3; k = 10 in both marker sets).
ALIGN3_OPTION_SETS = [
    dict(),                                                           # src/AssemblerOptions.cpp defaults
    dict(downsamplingFactor=0.05, minAlignedMarkerCount=10, minAlignedFraction=0.1, maxSkip=100, maxDrift=100,
         maxTrim=100, suppressContainments=1),                        # the shipped Nanopore confs
    dict(downsamplingFactor=0.3, bandExtend=3, maxBand=40, minAlignedMarkerCount=30),   # "band too wide" lane
]


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name))
        self.toc = self.z["toc"]
        kmer = self.z["kmer_ids"]
        pos = self.z["positions"]
        m = len(kmer)
        d = np.zeros((m, 7), np.uint8)
        d[:, 0:4] = np.ascontiguousarray(kmer, "<u4").view(np.uint8).reshape(m, 4)
        d[:, 4:7] = np.ascontiguousarray(pos, "<u4").view(np.uint8).reshape(m, 4)[:, 0:3]
        self.data7 = d.reshape(-1)
        self.kmer_ids = kmer

    def flags(self, i):
        return self.z["flags1"] if i == 1 else None

    def candidates(self, i=0):
        c = self.z["lh%d_candidates" % i]
        return abi.make_pairs(c[:, 0], c[:, 1], c[:, 2])


def check_lowhash(out, z, i):
    assert np.array_equal(out.candidate_tuples(), z["lh%d_candidates" % i])
    assert np.array_equal(out.statistics, z["lh%d_statistics" % i])
    assert np.array_equal(out.high_frequency, z["lh%d_high" % i])
    assert np.array_equal(out.total, z["lh%d_total" % i])
    assert np.array_equal(out.histogram, z["lh%d_histogram" % i])
    assert out.log2_bucket_count == int(z["lh%d_log2" % i][0])


def check_align(out, z, i):
    import hashlib
    assert np.array_equal(out.status & 0x7f, z["al%d_status" % i])
    assert np.array_equal(np.diff(out.ordinals_toc.astype(np.int64)), z["al%d_marker_count" % i])
    assert hashlib.md5(out.ordinals.tobytes()).hexdigest().encode() == z["al%d_ordinals_md5" % i].tobytes()
    assert np.array_equal(out.info_table(), z["al%d_info" % i])
    assert np.array_equal(out.compressed_toc, z["al%d_compressed_toc" % i])
    assert np.array_equal(out.compressed_data, z["al%d_compressed_data" % i])


def check_align3(out, z, i):
    import hashlib
    assert np.array_equal(out.status, z["m3_%d_status" % i])
    assert np.array_equal(np.diff(out.ordinals_toc.astype(np.int64)), z["m3_%d_marker_count" % i])
    assert hashlib.md5(out.ordinals.tobytes()).hexdigest().encode() == z["m3_%d_ordinals_md5" % i].tobytes()
    assert np.array_equal(out.info_table(), z["m3_%d_info" % i])
    assert np.array_equal(out.compressed_toc, z["m3_%d_compressed_toc" % i])
    assert np.array_equal(out.compressed_data, z["m3_%d_compressed_data" % i])


def same_lowhash(a, b):
    assert np.array_equal(a.candidate_tuples(), b.candidate_tuples())
    assert np.array_equal(a.statistics, b.statistics)
    assert np.array_equal(a.high_frequency, b.high_frequency)
    assert np.array_equal(a.total, b.total)
    assert np.array_equal(a.histogram, b.histogram)
    assert a.log2_bucket_count == b.log2_bucket_count


def _where_ordinals_differ(a, b):
    """For the assertion's message: the first candidate whose aligned pairs differ, how many values differ in all, the two lists there."""
    x, y = np.asarray(a.ordinals).reshape(-1), np.asarray(b.ordinals).reshape(-1)
    d = np.nonzero(x != y)[0]
    toc = np.asarray(a.ordinals_toc).astype(np.int64)
    k = int(np.searchsorted(toc, d[0] // 2, side="right") - 1)
    lo, hi = 2 * int(toc[k]), 2 * int(toc[k + 1])
    return "aligned pairs differ in %d values; first at candidate %d (pairs %d..%d): %s  vs  %s" % (len(d), k, lo // 2, hi // 2, x[lo:hi][:40].tolist(), y[lo:hi][:40].tolist())


def same_align(a, b, ties_ok=True):
    sa, sb = a.status & 0x7f, b.status & 0x7f
    assert np.array_equal(sa, sb)
    assert np.array_equal(a.status & 0x80, b.status & 0x80)
    if a.ordinals_toc is not None and b.ordinals_toc is not None:
        assert np.array_equal(a.ordinals_toc, b.ordinals_toc)
        assert np.array_equal(a.ordinals, b.ordinals), _where_ordinals_differ(a, b)
    assert np.array_equal(a.info_table(), b.info_table())
    assert np.array_equal(a.compressed_toc, b.compressed_toc)
    assert np.array_equal(a.compressed_data, b.compressed_data)


def small_marker_set(n_reads=300, genome_markers=20000, seed=5, **kw):
    toc, kmer = synthetic.marker_reads(n_reads, genome_markers, mean_markers=900.0, min_markers=300,
                                       seed=seed, **kw)
    return toc, kmer, synthetic.pack_markers(toc, kmer)


def reference_threads(limit=16):
    """Threads for the reference's aligner (oracle/_ref): every one of its threads creates the reference's own 2-GiB arena
    (src/AssemblerAlign.cpp:353-355), so "one per core" on a 256-core box is half a terabyte -- round 1 lost a GPU box that way.
    At most `limit`, the cores there are, the container's CPU quota, and a quarter of the available memory in 4-GiB units."""
    cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(float(quota) / float(period))))
    except Exception:          # noqa: BLE001
        pass
    memory = 1 << 20
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                memory = int(line.split()[1]) >> 20
    except OSError:
        pass
    return max(1, min(limit, cores, memory // 16))
