"""Parity checks of marker finding shared by the GPU tests and their emulated pre-flight."""
import os

import numpy as np

from shasta_amd import abi
from tests import support


def tiny_reads():
    z = np.load(os.path.join(support.GOLDEN, "tiny_reads.npz"))
    is_marker = np.unpackbits(z["is_marker_bits"], bitorder="little")[:1 << 20]
    return z["reads_toc"], z["reads_data"], z["base_counts"], is_marker


def golden_fixture(find_markers):
    """find_markers(reads_toc, reads_data, base_counts, k, is_marker) -> (toc, data7): the markers the
    reference's MarkerFinder made from the same reads (tiny.npz)."""
    rt, rd, bc, im = tiny_reads()
    g = support.Golden("tiny.npz")
    toc, data7 = find_markers(rt, rd, bc, 10, im)
    assert np.array_equal(toc, g.toc)
    assert np.array_equal(data7, g.data7)


def random_reads(seed, n_reads=30, k=10, fraction=0.12):
    """Random reads in LongBaseSequences layout, lengths around multiples of 64 and below k included."""
    rng = np.random.default_rng(seed)
    lengths = list(rng.integers(200, 3000, size=n_reads)) + [0, 1, k - 1, k, k + 1, 63, 64, 65, 127, 128, 129, 64 * 5 + k - 1]
    toc, words, counts = [0], [], []
    for n in lengths:
        n = int(n)
        bases = rng.integers(0, 4, size=n, dtype=np.uint8)
        blocks = (n + 63) // 64
        padded = np.zeros(blocks * 64, np.uint8)
        padded[:n] = bases
        low = np.packbits(padded & 1).view(">u8") if blocks else np.zeros(0, ">u8")
        high = np.packbits(padded >> 1).view(">u8") if blocks else np.zeros(0, ">u8")
        inter = np.empty(2 * blocks, np.uint64)
        inter[0::2] = low.astype(np.uint64)
        inter[1::2] = high.astype(np.uint64)
        words.append(inter)
        counts.append(n)
        toc.append(toc[-1] + 2 * blocks)
    is_marker = (rng.random(1 << (2 * k)) < fraction).astype(np.uint8)
    return (np.array(toc, np.uint64), np.concatenate(words) if words else np.zeros(0, np.uint64),
            np.array(counts, np.uint64), is_marker)


def against_oracle(lib, oracle_lib, seed, k):
    rt, rd, bc, im = random_reads(seed, k=k)
    a_toc, a_data = oracle_lib.find_markers(rt, rd, bc, k, im)
    b_toc, b_data = lib.find_markers(rt, rd, bc, k, im)
    assert np.array_equal(a_toc, b_toc) and np.array_equal(a_data, b_data)
    assert int(a_toc[-1]) > 100


def resident_markers_feed_lowhash0(lib):
    """find_markers on a context, then LowHash0 on the resident markers: the candidates the reference
    found from the markers of the same reads (tiny.npz, parameter set 0)."""
    rt, rd, bc, im = tiny_reads()
    g = support.Golden("tiny.npz")
    with lib.context(0) as ctx:
        toc, data7 = lib.find_markers(rt, rd, bc, 10, im, want_packed=False, context=ctx)
        assert data7 is None and np.array_equal(toc, g.toc)
        out = ctx.lowhash0(abi.default_lowhash0_params())
    support.check_lowhash(out, g.z, 0)
