"""The HIP path at benchmark scale (BASELINE configs[2] shape, 20 k reads, 6e7 markers, 4e5 candidates:
four Align4 batches, both host workers, every table class): too large for the oracle to recompute in
full, so the outputs are checked through the size-independent properties of tests/properties.py
(validated against the oracle in tests/test_properties_cpu.py), through idempotence, and exactly against
the oracle on a random sample of the candidates.  Named to run last among the GPU tests."""
import numpy as np
import pytest

import bench
from shasta_amd import abi, synthetic
from tests import properties, support

pytestmark = pytest.mark.gpu


def test_large_run_properties_and_sampled_parity(gpu_lib, oracle_lib):
    toc, kmer = bench.make_workload(20000, 4242)
    flags = np.zeros(20000, np.uint8)
    flags[::997] = 1
    p, o = bench.lowhash_params(), bench.align_options()
    with gpu_lib.context(0) as ctx:
        ctx.set_kmer_ids(toc, kmer, flags)
        lh = ctx.lowhash0(p)
        assert len(lh.candidates) > 200000
        properties.check_lowhash0(toc, flags, p, lh)
        again = ctx.lowhash0(p)
        support.same_lowhash(lh, again)                                   # idempotent on resident markers
        al = ctx.align4(lh.candidates, o, want_ordinals=False)
        borrowed = ctx.align4(lh.candidates, o, want_ordinals=False, borrow=True)
        assert np.array_equal(al.status, borrowed.status) and np.array_equal(al.compressed_data, borrowed.compressed_data)
        assert np.array_equal(al.info_table(), borrowed.info_table())
    # Skipped candidates (geometry beyond every kernel) and component ties (flagged, not resolved
    # silently) are legitimate but must be rare.
    assert ((al.status & 0x7f) == abi.SHASTA_ALIGN_SKIPPED).mean() < 1e-3 and ((al.status & 0x80) != 0).mean() < 1e-2
    assert len(al.alignment_data) > 100000
    properties.check_align4(toc, kmer, lh.candidates, o, al, oracle_lib.decompress, sample=500, seed=1)

    # Exact parity with the oracle on a random sample of the candidates (status, AlignmentInfo, blob).
    rng = np.random.default_rng(7)
    eligible = np.flatnonzero(((al.status & 0x80) == 0) & ((al.status & 0x7f) != abi.SHASTA_ALIGN_SKIPPED))
    pick = np.sort(rng.choice(eligible, size=600, replace=False))
    data7 = None
    sub_reads = np.unique(np.concatenate([lh.candidates["readId0"][pick], lh.candidates["readId1"][pick]]))
    # The oracle takes packed markers: build them for the reads of the sample only, renumbered.
    remap = {int(r): k for k, r in enumerate(sub_reads)}
    parts, sizes = [], []
    t64 = toc.astype(np.int64)
    for r in sub_reads:
        for strand in (0, 1):
            seg = kmer[t64[2 * r + strand]:t64[2 * r + strand + 1]]
            parts.append(seg); sizes.append(len(seg))
    sub_toc = np.zeros(len(sizes) + 1, np.uint64)
    sub_toc[1:] = np.cumsum(sizes)
    sub_kmer = np.concatenate(parts)
    data7 = synthetic.pack_markers(sub_toc, sub_kmer)
    c = lh.candidates[pick]
    sub_cand = abi.make_pairs([remap[int(x)] for x in c["readId0"]], [remap[int(x)] for x in c["readId1"]], c["isSameStrand"])
    ref = oracle_lib.align4_batch(sub_toc, data7, sub_cand, o, want_ordinals=False, threads=0)
    assert np.array_equal(ref.status & 0x7f, al.status[pick] & 0x7f)
    stored_index = np.cumsum((al.status & 0x7f) == abi.SHASTA_ALIGN_STORED) - 1
    k_ref = 0
    for i in pick:
        if (al.status[i] & 0x7f) != abi.SHASTA_ALIGN_STORED:
            continue
        k = int(stored_index[i])
        row, ref_row = al.alignment_data[k], ref.alignment_data[k_ref]
        for field in abi.INFO_FIELDS:
            assert row[field] == ref_row[field], (i, field)
        blob = al.compressed_data[int(al.compressed_toc[k]):int(al.compressed_toc[k + 1])]
        ref_blob = ref.compressed_data[int(ref.compressed_toc[k_ref]):int(ref.compressed_toc[k_ref + 1])]
        assert np.array_equal(blob, ref_blob), i
        k_ref += 1
    assert k_ref == len(ref.alignment_data) > 100
