"""The device side of SURVEY 8(f) row 3 (shasta_mi355x_pair_table, shasta_mi355x_read_graph_keep) against the python
restatements of computeAlignmentTable / computeCandidateTable / createReadGraph's selection in tests/host_support.py:
random pair lists with ties in every key (many pairs per oriented read, equal marker counts), shared by the GPU test
and its pre-flight on the emulated build."""
import numpy as np

from shasta_amd import abi
from tests import host_support


def random_alignment_data(rng, read_count, n):
    rows = np.zeros(n, dtype=abi.ALIGNMENT_DATA_DTYPE)
    a = rng.integers(0, read_count, size=n)
    b = rng.integers(0, read_count - 1, size=n)
    b = np.where(b >= a, b + 1, b)
    rows["readId0"] = np.minimum(a, b); rows["readId1"] = np.maximum(a, b)
    rows["isSameStrand"] = rng.integers(0, 2, size=n)
    rows["markerCount"] = rng.integers(90, 110, size=n)          # few distinct values: the alignment id breaks the ties
    return rows


def check(lib, seed=5, read_count=211, n=3000):
    rng = np.random.default_rng(seed)
    rows = random_alignment_data(rng, read_count, n)
    toc, values = lib.pair_table(rows, read_count)
    toc_expected, values_expected = host_support.alignment_table_expected(read_count, rows)
    assert np.array_equal(toc, toc_expected.astype(np.uint64)) and np.array_equal(values, values_expected)
    candidates = np.zeros(n, dtype=abi.PAIR_DTYPE)
    for name in ("readId0", "readId1", "isSameStrand"):
        candidates[name] = rows[name]
    toc12, values12 = lib.pair_table(candidates, read_count)
    assert np.array_equal(toc12, toc) and np.array_equal(values12, values)
    # The same tables built range by range (what a list of 2^30 pairs or more takes: one radix sort holds 2^31 entries), at a
    # size a test can hold: SHASTA_MI355X_PAIR_TABLE_KEYS = entries per sorted range, a quarter of it pairs per slab.  0.4 n = ten
    # ranges and ten slabs; 4 n - 1 = two ranges, one slab short of everything; n / 6 with every pair of one read = a single
    # oriented read beyond a range.
    import os
    import pytest
    try:
        for limit in (4 * n // 10 + 100, 4 * n - 1) + ((257,) if n <= 5000 else ()):
            os.environ["SHASTA_MI355X_PAIR_TABLE_KEYS"] = str(limit)
            for table in (rows, candidates):
                toc_r, values_r = lib.pair_table(table, read_count)
                assert np.array_equal(toc_r, toc) and np.array_equal(values_r, values), limit
            with pytest.raises(RuntimeError, match="beyond readCount"):
                lib.pair_table(rows, read_count // 2)
        os.environ["SHASTA_MI355X_PAIR_TABLE_KEYS"] = str(n // 6)
        crowded = candidates.copy()
        crowded["readId0"] = 0
        crowded["readId1"] = 1 + np.arange(n) % (read_count - 1)
        with pytest.raises(RuntimeError, match="more pairs than one sort takes"):
            lib.pair_table(crowded, read_count)
    finally:
        os.environ.pop("SHASTA_MI355X_PAIR_TABLE_KEYS", None)
    for k in (1, 6, 30, 10 ** 6):
        keep = lib.read_graph_keep(rows, read_count, k)
        expected = host_support.read_graph_expected(read_count, rows, k)[0]
        assert np.array_equal(keep.astype(bool), expected), k
    # Nothing to do, and what cannot be right.
    toc0, values0 = lib.pair_table(rows[:0], read_count)
    assert not toc0.any() and len(values0) == 0 and len(lib.read_graph_keep(rows[:0], read_count, 6)) == 0
    with pytest.raises(RuntimeError, match="beyond readCount"):
        lib.pair_table(rows, read_count // 2)
    with pytest.raises(RuntimeError, match="beyond readCount"):
        lib.read_graph_keep(rows, read_count // 2, 6)


def table_of_the_last_aligner_call(lib, oracle_lib, n_reads=150, limit=700):
    """shasta_mi355x_alignment_table: the last step of computeAlignments on the context that holds the alignments -- equal to
    the restatement of computeAlignmentTable on the rows the call returned, for both aligners, and refused without a call."""
    import pytest
    from tests import support
    toc, kmer, data7 = support.small_marker_set(n_reads=n_reads, genome_markers=12000, seed=77)
    cand = oracle_lib.lowhash0(toc, data7, None, abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30, minFrequency=1)).candidates[:limit]
    stored = 0
    with lib.context(0) as ctx:
        ctx.set_kmer_ids(toc, kmer)
        with pytest.raises(RuntimeError, match="holds no alignments"):
            ctx.alignment_table()
        for run, options in ((ctx.align4, abi.default_align4_options(minAlignedMarkerCount=40)), (ctx.align3, abi.default_align3_options(minAlignedMarkerCount=40)),
                             (ctx.align4, abi.default_align4_options(minAlignedMarkerCount=100000))):
            out = run(cand, options, borrow=True)
            rows = np.array(out.alignment_data, copy=True)
            expected_toc, expected_values = host_support.alignment_table_expected(n_reads, rows)
            for again in range(2):          # (the second time the keys the batches left on the device are gone: the rows go up from the host)
                table_toc, table_values = ctx.alignment_table()
                assert np.array_equal(table_toc, expected_toc.astype(np.uint64)) and np.array_equal(table_values, expected_values)
            stored += len(rows)
            del out
    return stored
