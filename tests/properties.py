"""Size-independent properties of the hot path's outputs (used where the oracle is too slow to
recompute everything: large GPU runs).  Every check follows from the reference's semantics:

LowHash0 (src/LowHash0.cpp):
  * candidates are strictly increasing in (readId0, readId1, strand) with readId0 < readId1 (:204-214, LowHash0.hpp:131-134);
  * "total" never decreases over the iterations and the last "high frequency" is the candidate count (:184-214);
  * per iteration, sum over histogram rows of size * count = number of low hashes of that iteration, and
    the row counts sum to 2^log2 buckets (:566-613); summed over iterations the low hashes equal the
    grand total of the per-read statistics (:386-393);
  * palindromic reads and reads shorter than m have all-zero statistics and appear in no candidate (:325,337).
Align4 + outer filters (src/AssemblerAlign.cpp:381-483, src/Alignment.cpp:67-113):
  * one status per candidate; rows / compressed blobs exist exactly for the STORED ones, in candidate order;
  * every stored row satisfies the acceptance filters it passed and its AlignmentInfo is internally consistent;
  * a stored alignment's compressed blob decodes (shasta::decompress) to markerCount strictly increasing ordinal
    pairs whose kmer ids are EQUAL in the two oriented reads, and whose first/last/min/max/skip/drift reproduce the row.
"""
import numpy as np

from shasta_amd import abi


def check_lowhash0(toc, flags, params, out):
    toc = np.asarray(toc, np.int64)
    read_count = (len(toc) - 1) // 2
    c = out.candidates
    r0, r1, same = c["readId0"].astype(np.int64), c["readId1"].astype(np.int64), c["isSameStrand"].astype(np.int64)
    assert np.all(r0 < r1) and (len(r1) == 0 or r1.max() < read_count)
    key = (r0 << 33) | (r1 << 1) | (1 - same)                   # strand 0 (= same strand) first
    assert np.all(np.diff(key) > 0)
    total = out.total.astype(np.int64)
    assert np.all(np.diff(total) >= 0)
    if len(out.high_frequency):
        assert int(out.high_frequency[-1]) == len(c)
        assert np.all(out.high_frequency <= out.total)
    h = out.histogram.astype(np.int64)
    low_hashes = 0
    for iteration in range(len(out.total)):
        rows = h[h[:, 0] == iteration]
        assert rows[:, 2].sum() == (1 << out.log2_bucket_count)
        assert np.all(np.diff(rows[:, 1]) > 0)
        low_hashes += int((rows[:, 1] * rows[:, 2]).sum())
    assert low_hashes == int(out.statistics.astype(np.int64).sum())
    sizes = toc[1::2] - toc[0:-1:2]
    silent = sizes < int(params.m)
    if flags is not None:
        silent |= (np.asarray(flags) & 1).astype(bool)
    assert not out.statistics[silent].any()
    assert not silent[r0].any() and not silent[r1].any()


def check_align4(toc, kmer_ids, candidates, options, out, decompress, sample=400, seed=0):
    toc = np.asarray(toc, np.int64)
    status = out.status & 0x7f
    assert len(status) == len(candidates)
    stored = np.flatnonzero(status == abi.SHASTA_ALIGN_STORED)
    rows = out.alignment_data
    assert len(rows) == len(stored) and len(out.compressed_toc) == len(rows) + 1
    assert np.array_equal(rows["readId0"], candidates["readId0"][stored])
    assert np.array_equal(rows["readId1"], candidates["readId1"][stored])
    assert np.array_equal(rows["isSameStrand"] != 0, candidates["isSameStrand"][stored] != 0)
    ctoc = out.compressed_toc.astype(np.int64)
    assert ctoc[0] == 0 and np.all(np.diff(ctoc) > 0) and ctoc[-1] == len(out.compressed_data)
    if len(rows) == 0:
        return
    # AlignmentInfo consistency and the filters of src/AssemblerAlign.cpp:439-472, for every row.
    n = rows["markerCount"].astype(np.int64)
    o0 = 2 * rows["readId0"].astype(np.int64)
    o1 = 2 * rows["readId1"].astype(np.int64) + (rows["isSameStrand"] == 0)
    assert np.array_equal(rows["markerCount0"], toc[o0 + 1] - toc[o0]) and np.array_equal(rows["markerCount1"], toc[o1 + 1] - toc[o1])
    range0 = rows["lastOrdinal0"].astype(np.int64) + 1 - rows["firstOrdinal0"]
    range1 = rows["lastOrdinal1"].astype(np.int64) + 1 - rows["firstOrdinal1"]
    assert np.all(n >= int(options.minAlignedMarkerCount)) and np.all(n <= np.minimum(range0, range1))
    assert np.all(np.minimum(n / range0, n / range1) >= options.minAlignedFraction)
    assert np.all(rows["maxSkip"] <= options.maxSkip) and np.all(rows["maxDrift"] <= options.maxDrift)
    left = np.minimum(rows["firstOrdinal0"], rows["firstOrdinal1"])
    right = np.minimum(rows["markerCount0"].astype(np.int64) - 1 - rows["lastOrdinal0"], rows["markerCount1"].astype(np.int64) - 1 - rows["lastOrdinal1"])
    assert np.all(left <= options.maxTrim) and np.all(right <= options.maxTrim)
    assert np.all(rows["minOrdinalOffset"] <= rows["averageOrdinalOffset"]) and np.all(rows["averageOrdinalOffset"] <= rows["maxOrdinalOffset"])
    # Decode a sample of the blobs and replay the metrics.
    rng = np.random.default_rng(seed)
    for k in rng.choice(len(rows), size=min(sample, len(rows)), replace=False):
        ordinals = decompress(out.compressed_data[ctoc[k]:ctoc[k + 1]]).astype(np.int64)
        x, y = ordinals[:, 0], ordinals[:, 1]
        row = rows[k]
        assert len(x) == row["markerCount"] and np.all(np.diff(x) > 0) and np.all(np.diff(y) > 0)
        a = kmer_ids[toc[o0[k]] + x]
        b = kmer_ids[toc[o1[k]] + y]
        assert np.array_equal(a, b)                                  # aligned markers have equal kmer ids
        assert (x[0], x[-1], y[0], y[-1]) == (row["firstOrdinal0"], row["lastOrdinal0"], row["firstOrdinal1"], row["lastOrdinal1"])
        offset = x - y
        assert offset.min() == row["minOrdinalOffset"] and offset.max() == row["maxOrdinalOffset"]
        assert int(np.round(offset.sum() / float(len(x)))) == row["averageOrdinalOffset"] or \
            abs(offset.sum() / float(len(x)) - row["averageOrdinalOffset"]) <= 0.5 + 1e-9
        if len(x) > 1:
            assert max(np.diff(x).max(), np.diff(y).max()) == row["maxSkip"]
            assert np.abs(np.diff(offset)).max() == row["maxDrift"]
