"""More than 2^32 markers on one GPU (17.8 GB of dense kmer ids: SURVEY 8d's human-scale shapes have 64-bit marker offsets,
src/Marker.hpp:44-48, and nothing smaller crosses that line): 740 000 noisy reads of 3000 markers generated ON the device,
LowHash0 over all of them -- both code paths, all iterations in one pass and iteration after iteration, must agree and satisfy
what can be checked without recomputing (order, true overlaps only, every large overlap found) -- and Align4 on candidates
whose reads lie beyond offset 2^32, bit for bit against the oracle on those reads copied out.  Named to run last."""
import numpy as np
import pytest

from shasta_amd import abi, synthetic
from tests import support

pytestmark = pytest.mark.gpu

READS, LENGTH, KEEP, COVERAGE = 740_000, 3300, 0.9, 12


def generate_on_device(torch, device):
    """Reads = windows of a random genome with one marker in ten dropped; strand 1 = reversed with the low bit flipped (the
    involution tests/adversarial.py uses in place of the reverse complement).  Returns toc (host), the flat kmer ids
    (device tensor, read-major, strand 0 then strand 1) and the window starts (host)."""
    g = torch.Generator(device=device)
    g.manual_seed(20260927)
    genome_markers = READS * int(LENGTH * KEEP) // COVERAGE
    genome = torch.randint(0, 1 << 20, (genome_markers + LENGTH,), dtype=torch.int32, device=device, generator=g)
    starts = torch.randint(0, genome_markers, (READS,), dtype=torch.int64, device=device, generator=g)
    lengths = torch.empty(READS, dtype=torch.int64, device=device)
    chunk = 20000
    masks = []
    for r0 in range(0, READS, chunk):                       # pass 1: which markers stay (the masks are kept as bits: 2.4e9 / 8 bytes)
        n = min(chunk, READS - r0)
        mask = torch.rand((n, LENGTH), device=device, generator=g) < KEEP
        lengths[r0:r0 + n] = mask.sum(dim=1)
        masks.append(mask)
    toc = torch.zeros(2 * READS + 1, dtype=torch.int64, device=device)
    toc[1:] = torch.cumsum(torch.repeat_interleave(lengths, 2), dim=0)
    total = int(toc[-1].item())
    flat = torch.empty(total, dtype=torch.int32, device=device)
    column = torch.arange(LENGTH, device=device)
    for k, r0 in enumerate(range(0, READS, chunk)):
        n = min(chunk, READS - r0)
        mask = masks[k]
        full = genome[(starts[r0:r0 + n, None] + column[None, :])]
        base0 = toc[2 * r0:2 * (r0 + n):2, None]
        rank = torch.cumsum(mask, dim=1) - 1
        flat[(base0 + rank)[mask]] = full[mask]
        reversed_mask = mask.flip(1)
        base1 = toc[2 * r0 + 1:2 * (r0 + n) + 1:2, None]
        rank1 = torch.cumsum(reversed_mask, dim=1) - 1
        flat[(base1 + rank1)[reversed_mask]] = (full.flip(1) ^ 1)[reversed_mask]
        masks[k] = None
    return toc.cpu().numpy().astype(np.uint64), flat, starts.cpu().numpy()


def test_lowhash0_and_align4_with_marker_offsets_beyond_32_bits(gpu_lib, oracle_lib, monkeypatch):
    torch = pytest.importorskip("torch")
    if gpu_lib.path.endswith("_emu.so"):
        pytest.skip("18 GB of markers: the MI355X only")
    device = torch.device("cuda", 0)
    toc, flat, starts = generate_on_device(torch, device)
    assert int(toc[-1]) > (1 << 32) + (1 << 26)
    p = abi.default_lowhash0_params(minHashIterationCount=4, minBucketSize=5, maxBucketSize=30, minFrequency=2)
    o = abi.default_align4_options(minAlignedMarkerCount=100)
    with gpu_lib.context(0) as ctx:
        torch.cuda.synchronize()
        ctx.set_kmer_ids_device(toc, flat.data_ptr())
        lh = ctx.lowhash0(p)
        cand = np.array(lh.candidates)                      # (the result's buffer is the library's)
        stats, high = np.array(lh.statistics), np.array(lh.high_frequency)
        monkeypatch.setenv("SHASTA_MI355X_LOWHASH_ONE_PASS", "0")
        again = ctx.lowhash0(p)
        monkeypatch.delenv("SHASTA_MI355X_LOWHASH_ONE_PASS")
        assert np.array_equal(cand, np.array(again.candidates)) and np.array_equal(stats, again.statistics) and np.array_equal(high, again.high_frequency)
        del again
        # Order (src/LowHash0.hpp:131-134), and only true overlaps: the reads are windows of one random genome, so two reads share
        # a window of four markers only where their windows of the genome overlap, on the same strand.
        r0, r1, same = cand["readId0"].astype(np.int64), cand["readId1"].astype(np.int64), cand["isSameStrand"].astype(np.int64)
        assert len(cand) > 2_000_000 and (r0 < r1).all()
        key = (r0 << 22) | (r1 << 1) | (1 - same)
        assert (np.diff(key) > 0).all()
        assert (same == 1).all() and (np.abs(starts[r0] - starts[r1]) < LENGTH).all()
        # Every pair of reads that share half of their windows is a candidate (about 26 common low hashes expected): the reads
        # whose window starts are among the 20 000 smallest, all pairs.
        order = np.argsort(starts, kind="stable")[:20000]
        s = starts[order]
        expected = set()
        for a in range(len(order)):
            b = a + 1
            while b < len(order) and s[b] - s[a] <= LENGTH // 2:
                x, y = int(order[a]), int(order[b])
                expected.add((min(x, y), max(x, y)))
                b += 1
        sub = np.isin(r0, order) & np.isin(r1, order)
        found = set(zip(r0[sub].tolist(), r1[sub].tolist()))
        assert len(expected) > 50000 and expected <= found
        # Align4 on candidates whose BOTH reads lie beyond offset 2^32.
        beyond = np.flatnonzero(toc[2 * r0] >= (1 << 32))
        assert len(beyond) > 3000
        pick = beyond[np.linspace(0, len(beyond) - 1, 1500).astype(np.int64)]
        c = cand[pick]
        al = ctx.align4(c, o, want_ordinals=True)
        # The reads of the sample, copied out and renumbered for the oracle.
        sub_reads = np.unique(np.concatenate([c["readId0"], c["readId1"]]))
        remap = {int(r): k for k, r in enumerate(sub_reads)}
        parts, sizes = [], []
        for r in sub_reads:
            begin, end = int(toc[2 * r]), int(toc[2 * r + 2])
            both = flat[begin:end].cpu().numpy().view(np.uint32)
            n0 = int(toc[2 * r + 1]) - begin
            parts += [both[:n0], both[n0:]]; sizes += [n0, len(both) - n0]
    sub_toc = np.zeros(len(sizes) + 1, np.uint64)
    sub_toc[1:] = np.cumsum(sizes)
    data7 = synthetic.pack_markers(sub_toc, np.concatenate(parts))
    sub_cand = abi.make_pairs([remap[int(x)] for x in c["readId0"]], [remap[int(x)] for x in c["readId1"]], c["isSameStrand"])
    ref = oracle_lib.align4_batch(sub_toc, data7, sub_cand, o, want_ordinals=True, threads=0)
    assert ((ref.status & 0x7f) == abi.SHASTA_ALIGN_STORED).sum() > 1000
    assert np.array_equal(ref.status, al.status) and np.array_equal(ref.ordinals_toc, al.ordinals_toc) and np.array_equal(ref.ordinals, al.ordinals)
    assert np.array_equal(ref.compressed_toc, al.compressed_toc) and np.array_equal(ref.compressed_data, al.compressed_data)
    assert np.array_equal(ref.info_table()[:, 2:], al.info_table()[:, 2:])           # (read ids renumbered; everything else equal)
