"""Seeded synthetic inputs for tests and bench.py.

Two generators:

* ``fasta_reads``   base-level ONT-like reads (random genome, lognormal lengths, random strand,
                    substitution/insertion/deletion errors).  Used only to make fixtures through
                    the reference's own ReadLoader -> MarkerFinder (tests/golden/make_golden.py).
* ``marker_reads``  marker-level reads: the hot path never touches bases (LowHash0 and
                    Align4 read only ``markers[*].kmerId``), so large workloads are generated
                    directly as marker sequences: a random "genome" of marker k-mer ids, reads =
                    noisy substrings (marker loss, spurious markers), strand 1 = reversed sequence
                    under the reverse-complement involution, exactly the relation MarkerFinder
                    produces (src/MarkerFinder.cpp:98-110).
"""
import numpy as np


def reverse_complement_ids(ids, k):
    """Reverse complement of 2-bit packed k-mers (A=0,C=1,G=2,T=3, first base most significant)."""
    ids = np.asarray(ids, dtype=np.uint64)
    out = np.zeros_like(ids)
    x = ids.copy()
    for _ in range(k):
        out = (out << np.uint64(2)) | (np.uint64(3) - (x & np.uint64(3)))
        x >>= np.uint64(2)
    return out


def marker_alphabet(k=10, probability=0.1, seed=231):
    """RLE k-mers (no equal adjacent bases) selected as markers, closed under reverse complement.
    Returns (marker_ids sorted uint32, rc_table uint32[4**k])."""
    n = 1 << (2 * k)
    ids = np.arange(n, dtype=np.uint64)
    rc = reverse_complement_ids(ids, k)
    is_rle = np.ones(n, dtype=bool)
    for i in range(1, k):
        a = (ids >> np.uint64(2 * (i - 1))) & np.uint64(3)
        b = (ids >> np.uint64(2 * i)) & np.uint64(3)
        is_rle &= a != b
    rng = np.random.default_rng(seed)
    p = 1.0 - np.sqrt(1.0 - probability)
    pick = rng.random(n) <= p
    marker = np.zeros(n, dtype=bool)
    marker[pick] = True
    marker[rc[pick].astype(np.int64)] = True
    marker &= is_rle
    return np.nonzero(marker)[0].astype(np.uint32), rc.astype(np.uint32)


class RcMap:
    """Reverse complement of the ids of a sampled alphabet: map[ids] for ids of the alphabet (an array lookup by binary
    search, where a table over all 4^k ids would be gigabytes at k = 14 and 16)."""
    def __init__(self, ids, rc):
        self.ids, self.rc = ids, rc

    def __getitem__(self, x):
        x = np.asarray(x)
        at = np.searchsorted(self.ids, x)
        assert np.array_equal(self.ids[np.minimum(at, len(self.ids) - 1)], x), "id outside the sampled alphabet"
        return self.rc[at]


def sampled_marker_alphabet(k=14, count=1 << 21, seed=231):
    """About `count` distinct RLE k-mers (no equal adjacent bases) drawn over the WHOLE id range of k (4^14 = 2^28 for the
    Nanopore-May2022 configurations, 4^16 = 2^32 at the largest k Shasta's 32-bit KmerId holds), closed under reverse
    complement -> (ids sorted uint32, RcMap).  marker_alphabet() enumerates all 4^k ids, which stops being practical beyond
    k = 12; the hot path only ever sees the ids, so a sample with the same structure exercises the same range."""
    rng = np.random.default_rng([seed, k])
    first = rng.integers(0, 4, size=count, dtype=np.uint64)
    steps = rng.integers(1, 4, size=(count, k - 1), dtype=np.uint64)
    bases = (first[:, None] + np.concatenate([np.zeros((count, 1), np.uint64), np.cumsum(steps, axis=1, dtype=np.uint64)], axis=1)) & np.uint64(3)
    ids = np.zeros(count, dtype=np.uint64)
    for i in range(k):
        ids = (ids << np.uint64(2)) | bases[:, i]
    ids = np.unique(np.concatenate([ids, reverse_complement_ids(ids, k)]))
    return ids.astype(np.uint32), RcMap(ids.astype(np.uint32), reverse_complement_ids(ids, k).astype(np.uint32))


def marker_genome(rng, alphabet, genome_markers, repeat_fraction=0.02):
    """A random genome over the marker alphabet with a few 200-marker repeats, as real genomes have."""
    genome = alphabet[rng.integers(0, len(alphabet), size=genome_markers)]
    n_rep = int(repeat_fraction * genome_markers / 200)
    for _ in range(n_rep):
        a = int(rng.integers(0, genome_markers - 200))
        b = int(rng.integers(0, genome_markers - 200))
        genome[b:b + 200] = genome[a:a + 200]
    return genome


def marker_reads(n_reads, genome_markers, mean_markers=1500.0, sigma=0.5, min_markers=700,
                 keep_probability=0.65, spurious_probability=0.06, k=10, seed=12345,
                 repeat_fraction=0.02, shard=None, shard_count=1, alphabet=None):
    """Returns (toc uint64[2R+1], kmer_ids uint32[M]) for R = n_reads reads, both strands.
    With shard / shard_count the genome (from `seed`) is common to all shards and the n_reads reads
    of shard `shard` come from their own stream, so that shards generated on different ranks are
    consecutive read ranges of one read set.  `alphabet` = (ids, reverse-complement lookup), e.g. sampled_marker_alphabet(14);
    default: all marker k-mers of marker_alphabet(k)."""
    rng = np.random.default_rng(seed)
    alphabet, rc_table = marker_alphabet(k=k) if alphabet is None else alphabet
    genome = marker_genome(rng, alphabet, genome_markers, repeat_fraction)
    if shard is not None:
        rng = np.random.default_rng([seed, 1000003 + int(shard), int(shard_count)])

    mu = np.log(mean_markers) - 0.5 * sigma * sigma
    span = np.maximum(min_markers, rng.lognormal(mu, sigma, size=n_reads)).astype(np.int64)
    span = np.minimum(span, genome_markers)
    start = (rng.random(n_reads) * (genome_markers - span + 1)).astype(np.int64)
    flip = rng.random(n_reads) < 0.5

    total = int(span.sum())
    read_of = np.repeat(np.arange(n_reads, dtype=np.int64), span)
    first = np.cumsum(span) - span
    offset = np.arange(total, dtype=np.int64) - np.repeat(first, span)
    gi = np.repeat(start, span) + offset
    keep = rng.random(total) < keep_probability
    read_of = read_of[keep]
    ids = genome[gi[keep]]
    del gi, offset, keep
    # Spurious markers: duplicate an element and overwrite the copy with a random marker.
    extra = rng.random(len(ids)) < spurious_probability
    reps = 1 + extra.astype(np.int64)
    idx = np.repeat(np.arange(len(ids), dtype=np.int64), reps)
    is_copy = np.zeros(len(idx), dtype=bool)
    is_copy[1:] = idx[1:] == idx[:-1]
    ids = ids[idx]
    read_of = read_of[idx]
    n_copy = int(is_copy.sum())
    ids[is_copy] = alphabet[rng.integers(0, len(alphabet), size=n_copy)]
    del idx, is_copy, extra, reps

    counts = np.bincount(read_of, minlength=n_reads).astype(np.int64)
    begin0 = np.cumsum(counts) - counts                     # start of each read in `ids`
    p = np.arange(len(ids), dtype=np.int64) - begin0[read_of]
    # Strand 0 as stored = the sequenced strand: flipped reads are reversed + reverse complemented.
    f = flip[read_of]
    src_rev = begin0[read_of] + (counts[read_of] - 1 - p)
    strand0 = np.where(f, rc_table[ids[src_rev]], ids).astype(np.uint32)
    strand1 = rc_table[strand0[src_rev]].astype(np.uint32)  # reverse + RC of strand 0
    del f, src_rev

    toc = np.zeros(2 * n_reads + 1, dtype=np.uint64)
    sizes = np.repeat(counts, 2)
    toc[1:] = np.cumsum(sizes)
    out = np.empty(int(toc[-1]), dtype=np.uint32)
    dst0 = toc[0:-1:2].astype(np.int64)[read_of] + p
    out[dst0] = strand0
    out[dst0 + counts[read_of]] = strand1
    return toc, out


def pack_markers(toc, kmer_ids):
    """kmer ids -> packed 7-byte CompressedMarker records (u32 kmerId, u24 position).
    position = 10 * ordinal (monotone within a read; never read by the hot path)."""
    m = len(kmer_ids)
    out = np.zeros((m, 7), dtype=np.uint8)
    out[:, 0:4] = np.ascontiguousarray(kmer_ids, dtype="<u4").view(np.uint8).reshape(m, 4)
    toc = np.asarray(toc, dtype=np.int64)
    sizes = np.diff(toc)
    ordinal = np.arange(m, dtype=np.int64) - np.repeat(toc[:-1], sizes)
    pos = (10 * ordinal).astype("<u4") & np.uint32(0xFFFFFF)
    out[:, 4:7] = pos.view(np.uint8).reshape(m, 4)[:, 0:3]
    return out.reshape(-1)


def unpack_kmer_ids(data7):
    d = np.ascontiguousarray(data7, dtype=np.uint8).reshape(-1, 7)
    return np.ascontiguousarray(d[:, 0:4]).view("<u4").reshape(-1).copy()


def fasta_reads(path, n_reads, genome_length, mean_length=15000.0, sigma=0.35, min_length=10500,
                sub=0.02, ins=0.015, dele=0.015, seed=12345, homopolymer=0.0):
    """Writes base-level reads to a FASTA file; returns the list of (start, length, flipped).
    homopolymer: probability per base of a run-length error (the base is written twice, or dropped when it repeats its
    predecessor) -- the dominant Nanopore error, which Shasta's run-length representation removes again."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=genome_length, dtype=np.uint8)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    mu = np.log(mean_length) - 0.5 * sigma * sigma
    meta = []
    with open(path, "wb") as f:
        for r in range(n_reads):
            length = int(min(genome_length, max(min_length, rng.lognormal(mu, sigma))))
            start = int(rng.integers(0, genome_length - length + 1))
            s = genome[start:start + length].copy()
            u = rng.random(length)
            is_sub = u < sub
            s[is_sub] = (s[is_sub] + rng.integers(1, 4, size=int(is_sub.sum()), dtype=np.uint8)) % 4
            keep = ~((u >= sub) & (u < sub + dele))
            s = s[keep]
            is_ins = rng.random(len(s)) < ins
            reps = 1 + is_ins.astype(np.int64)
            idx = np.repeat(np.arange(len(s)), reps)
            copy = np.zeros(len(idx), dtype=bool)
            copy[1:] = idx[1:] == idx[:-1]
            s = s[idx]
            s[copy] = rng.integers(0, 4, size=int(copy.sum()), dtype=np.uint8)
            if homopolymer > 0.0:
                u = rng.random(len(s))
                repeats_previous = np.zeros(len(s), dtype=bool)
                repeats_previous[1:] = s[1:] == s[:-1]
                s = np.repeat(s, np.where(u < homopolymer, 2, np.where((u > 1.0 - homopolymer) & repeats_previous, 0, 1)))
            flipped = bool(rng.random() < 0.5)
            if flipped:
                s = comp[s[::-1]]
            f.write(b">read%d\n" % r)
            f.write(letters[s].tobytes())
            f.write(b"\n")
            meta.append((start, length, flipped))
    return meta


def packed_base_reads(n_reads, mean_bases=20000.0, sigma=0.5, min_bases=10500, seed=12345):
    """Random run-length-encoded reads (no two equal adjacent bases, as Shasta stores reads in RLE space) in the layout of
    Shasta's LongBaseSequences (src/LongBaseSequence.hpp:33-41): per read, per 64 bases two 64-bit words -- the low bits and the
    high bits of the bases, first base in the most significant bit.  -> (reads_toc uint64[R+1] in words, reads_data uint64[],
    base_counts uint64[R]).  For the marker-finding bench line; parity of that stage is tested on reference-made fixtures."""
    rng = np.random.default_rng(seed)
    mu = np.log(mean_bases) - 0.5 * sigma * sigma
    lengths = np.maximum(min_bases, rng.lognormal(mu, sigma, size=n_reads)).astype(np.int64)
    blocks = (lengths + 63) // 64
    toc = np.zeros(n_reads + 1, dtype=np.uint64)
    toc[1:] = np.cumsum(2 * blocks)
    total_blocks = int(blocks.sum())
    # One long RLE sequence cut into reads: base[i+1] = base[i] + 1..3 mod 4.
    steps = rng.integers(1, 4, size=total_blocks * 64, dtype=np.uint8)
    bases = (np.cumsum(steps, dtype=np.uint64) & np.uint64(3)).astype(np.uint8)
    # Bases past the end of a read (padding of its last block) are zero.
    block_start = np.concatenate([[0], np.cumsum(blocks)])[:-1] * 64
    index = np.arange(total_blocks * 64, dtype=np.int64)
    read_of_block = np.repeat(np.arange(n_reads), blocks)
    read_of_base = np.repeat(read_of_block, 64)
    bases[index - block_start[read_of_base] >= lengths[read_of_base]] = 0
    low = np.packbits(bases & 1).view(">u8").astype(np.uint64)
    high = np.packbits(bases >> 1).view(">u8").astype(np.uint64)
    data = np.empty(2 * total_blocks, dtype=np.uint64)
    data[0::2] = low
    data[1::2] = high
    return toc, data, lengths.astype(np.uint64)
