"""Palindromic-read flagging without a GPU: the restated method 0 against the
reference's own AlignmentGraph.cpp (oracle/_ref) and against the fixture it made; the product's host
half (shasta_amd/host/PalindromicReads.cpp) against the oracle, alignment by alignment; the device
screen and the whole stage on the emulated build."""
import os

import numpy as np
import pytest

from tests import host_support, palindromic_checks as pc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_SO = os.path.join(ROOT, "shasta_amd", "_build", "libshasta_mi355x_host.so")


def test_oracle_reproduces_the_reference_fixture(oracle_lib):
    pc.check_against_golden(oracle_lib.flag_palindromic_reads)


@pytest.mark.parametrize("seed", [21, 22])
def test_oracle_equals_reference(oracle_lib, ref_lib, seed):
    toc, kmer, data7, kinds = pc.read_set(n_reads=48, seed=seed)
    for kw in pc.PARAMETER_SETS:
        a = ref_lib.flag_palindromic_reads(toc, data7, threads=2, **kw)
        b = oracle_lib.flag_palindromic_reads(toc, data7, **kw)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_host_method0_equals_oracle_alignment_by_alignment(oracle_lib):
    # The product's host half on its own: same ordinals (through the digest), not only the same counts.
    host = pc.HostLib(HOST_SO)
    toc, kmer, data7, kinds = pc.read_set(n_reads=40, seed=31)
    for kw in pc.PARAMETER_SETS:
        flags, aligned, near, digests = oracle_lib.flag_palindromic_reads(toc, data7, **kw)
        for r in range(len(flags)):
            k0 = kmer[int(toc[2 * r]):int(toc[2 * r + 1])]
            k1 = kmer[int(toc[2 * r + 1]):int(toc[2 * r + 2])]
            a = host.self_alignment(k0, k1, kw["max_skip"], kw["max_drift"], kw["max_marker_frequency"])
            assert pc.counts_of(a, kw["delta_threshold"]) == (int(aligned[r]), int(near[r])), (r, kinds[r])
            h = 1469598103934665603
            for v in a.reshape(-1):
                for b in range(4):
                    h = ((h ^ ((int(v) >> (8 * b)) & 0xff)) * 1099511628211) & 0xffffffffffffffff
            assert h == int(digests[r]), (r, kinds[r])


def test_emulated_screen_is_the_bound_and_is_sound(emu_lib):
    z, inputs = pc.golden()
    toc, data7 = inputs["hairpins"]
    kmer = np.ascontiguousarray(data7.reshape(-1, 7)[:, :4]).view("<u4").reshape(-1)
    with emu_lib.context(0) as ctx:
        ctx.set_markers(toc, data7)
        for i, kw in enumerate(pc.PARAMETER_SETS):
            bound = ctx.palindromic_screen(kw["delta_threshold"])
            assert np.array_equal(bound, pc.numpy_bound(toc, kmer, kw["delta_threshold"]))
            screened = pc.screen_is_sound(bound, toc, z["hairpins_%d_near" % i], kw["near_diagonal_fraction_threshold"])
            assert not np.any(screened & (z["hairpins_%d_flags" % i] != 0))
            assert screened.sum() > 10                              # the screen settles the ordinary reads
        with pytest.raises(RuntimeError):
            ctx.palindromic_screen(0)
        with pytest.raises(RuntimeError):
            ctx.palindromic_screen(5000)


def test_emulated_stage_equals_reference_fixture(emu_lib, tmp_path):
    host = os.path.join(os.path.dirname(emu_lib.path), "libshasta_mi355x_host_emu.so")
    z, inputs = pc.golden()
    for name, (toc, data7) in inputs.items():
        for i, kw in enumerate(pc.PARAMETER_SETS):
            read_count = (len(toc) - 1) // 2
            before = (np.arange(read_count) % 4).astype(np.uint8)   # bit 0 set on some reads, bit 1 (another flag) on others
            flags, counts = pc.flag_through_stage(toc, data7, tmp_path / ("%s%d" % (name, i)), host, initial_flags=before, **kw)
            assert np.array_equal(flags & 1, z["%s_%d_flags" % (name, i)])
            assert np.array_equal(flags & 0xfe, before & 0xfe)      # only the palindromic bit is touched
            assert counts[0] == read_count and counts[2] == int(z["%s_%d_flags" % (name, i)].sum())
            if name == "hairpins":
                assert counts[1] > 10


EDGE_PARAMETER_SETS = (
    dict(max_skip=100, max_drift=100, max_marker_frequency=1, aligned_fraction_threshold=0.1, near_diagonal_fraction_threshold=0.1, delta_threshold=100),
    dict(max_skip=5, max_drift=200, max_marker_frequency=10, aligned_fraction_threshold=0.0, near_diagonal_fraction_threshold=0.0, delta_threshold=1),     # no drift test (maxDrift >= maxSkip); thresholds 0: every read is flagged
    dict(max_skip=1, max_drift=0, max_marker_frequency=2, aligned_fraction_threshold=0.5, near_diagonal_fraction_threshold=0.5, delta_threshold=3),
    dict(max_skip=1000, max_drift=3, max_marker_frequency=1000, aligned_fraction_threshold=1.0, near_diagonal_fraction_threshold=1.0, delta_threshold=4096),   # nothing is high-frequency
)


def degenerate_reads():
    """Reads that stress the tie rules: one kmer repeated, two kmers alternating, a perfect palindrome, single markers."""
    from shasta_amd import synthetic
    alphabet, rc = synthetic.marker_alphabet(k=10)
    a, b = np.uint32(alphabet[5]), np.uint32(alphabet[77])
    self_rc = [x for x in alphabet[:20000] if rc[x] == x][:3]            # kmers equal to their own reverse complement, if any
    strands0 = [np.full(40, a, np.uint32), np.tile(np.array([a, rc[a]], np.uint32), 30), np.tile(np.array([a, b], np.uint32), 25),
                np.array([a], np.uint32), np.array([a, rc[a]], np.uint32), np.zeros(0, np.uint32)]
    rng = np.random.default_rng(3)
    half = alphabet[rng.integers(0, len(alphabet), size=150)].astype(np.uint32)
    strands0.append(np.concatenate([half, rc[half[::-1]].astype(np.uint32)]))                   # exact palindrome
    if self_rc:
        strands0.append(np.array(self_rc * 10, np.uint32))
    sizes = np.repeat(np.asarray([len(s) for s in strands0], np.uint64), 2)
    toc = np.zeros(2 * len(strands0) + 1, np.uint64)
    toc[1:] = np.cumsum(sizes)
    kmer = np.concatenate([np.concatenate([s, rc[s[::-1]].astype(np.uint32)]) for s in strands0]).astype(np.uint32)
    return toc, kmer, synthetic.pack_markers(toc, kmer)


@pytest.mark.parametrize("which", ["mixed", "degenerate"])
def test_edge_parameters_oracle_reference_and_host_agree(oracle_lib, ref_lib, which):
    host = pc.HostLib(HOST_SO)
    if which == "mixed":
        toc, kmer, data7, _ = pc.read_set(n_reads=32, seed=41)
    else:
        toc, kmer, data7 = degenerate_reads()
    flagged = 0
    for kw in EDGE_PARAMETER_SETS:
        if which == "mixed" and kw["max_marker_frequency"] > 100:
            # every kmer of a low-complexity read becomes a vertex streak: graphs of 10^5 vertices with 10^3 neighbours each
            kw = dict(kw, max_marker_frequency=12, max_skip=150)
        a = ref_lib.flag_palindromic_reads(toc, data7, threads=2, **kw)
        b = oracle_lib.flag_palindromic_reads(toc, data7, **kw)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), kw
        flagged += int(a[0].sum())
        for r in range(len(a[0])):
            k0 = kmer[int(toc[2 * r]):int(toc[2 * r + 1])]
            k1 = kmer[int(toc[2 * r + 1]):int(toc[2 * r + 2])]
            got = host.self_alignment(k0, k1, kw["max_skip"], kw["max_drift"], kw["max_marker_frequency"])
            assert pc.counts_of(got, kw["delta_threshold"]) == (int(a[1][r]), int(a[2][r])), (kw, r)
    assert flagged > 0


def test_emulated_stage_with_edge_parameters(emu_lib, oracle_lib, tmp_path):
    host = os.path.join(os.path.dirname(emu_lib.path), "libshasta_mi355x_host_emu.so")
    toc, kmer, data7 = degenerate_reads()
    for i, kw in enumerate(EDGE_PARAMETER_SETS):
        expected = oracle_lib.flag_palindromic_reads(toc, data7, **kw)[0]
        flags, counts = pc.flag_through_stage(toc, data7, tmp_path / str(i), host, **kw)
        assert np.array_equal(flags & 1, expected), kw
