"""Inputs and checks for palindromic-read flagging, shared by the CPU tests (oracle,
reference, emulated build) and the -m gpu tests.  Test infrastructure."""
import ctypes as C

import numpy as np

from shasta_amd import synthetic

DEFAULTS = dict(max_skip=100, max_drift=100, max_marker_frequency=10,
                aligned_fraction_threshold=0.1, near_diagonal_fraction_threshold=0.1, delta_threshold=100)


def read_set(n_reads=60, seed=7, k=10):
    """Marker-level reads (both strands, strand 1 = reverse + reverse complement of strand 0) of several kinds:
    ordinary; palindromic (second half = noisy reverse complement of the first half, as a chimeric
    hairpin read is); partly palindromic around the thresholds; low complexity (a short unit and its
    reverse complement repeated: kmer streaks beyond maxMarkerFrequency); very short; empty.
    Returns (toc, kmer ids, packed 7-byte markers, kind per read)."""
    rng = np.random.default_rng(seed)
    alphabet, rc = synthetic.marker_alphabet(k=k)

    def noisy(x, loss, sub):
        x = x[rng.random(len(x)) >= loss].copy()
        s = rng.random(len(x)) < sub
        x[s] = alphabet[rng.integers(0, len(alphabet), size=int(s.sum()))]
        return x

    def revcomp(x):
        return rc[x[::-1]].astype(np.uint32)

    strands0, kinds = [], []
    for r in range(n_reads):
        kind = ("ordinary", "palindromic", "partial", "ordinary", "lowcomplexity", "palindromic", "short", "partial")[r % 8]
        if r == n_reads - 1:
            kind = "empty"
        n = int(rng.integers(300, 1400))
        base = alphabet[rng.integers(0, len(alphabet), size=n)].astype(np.uint32)
        if kind == "ordinary":
            s0 = base
        elif kind == "palindromic":
            half = base[:n // 2]
            s0 = np.concatenate([half, noisy(revcomp(half), 0.1, 0.1)])
        elif kind == "partial":
            # a hairpin in the middle covering a fraction of the read near the thresholds
            f = float(rng.choice([0.04, 0.08, 0.1, 0.12, 0.2, 0.4]))
            h = max(2, int(f * n / 2))
            arm = base[:h]
            left = alphabet[rng.integers(0, len(alphabet), size=(n - 2 * h) // 2)].astype(np.uint32)
            s0 = np.concatenate([left, arm, noisy(revcomp(arm), 0.05, 0.05), left[::-1].copy()])
        elif kind == "lowcomplexity":
            unit = base[:int(rng.integers(3, 9))]
            rep = np.concatenate([unit, revcomp(unit)])
            s0 = noisy(np.tile(rep, n // len(rep) + 1)[:n], 0.03, 0.03)
        elif kind == "short":
            s0 = base[:int(rng.integers(1, 12))]
            if r % 16 == 6:
                s0 = np.concatenate([s0, revcomp(s0)])
        else:
            s0 = base[:0]
        strands0.append(s0.astype(np.uint32))
        kinds.append(kind)
    sizes = np.repeat(np.asarray([len(s) for s in strands0], dtype=np.uint64), 2)
    toc = np.zeros(2 * n_reads + 1, dtype=np.uint64)
    toc[1:] = np.cumsum(sizes)
    kmer = np.concatenate([np.concatenate([s, revcomp(s)]) for s in strands0]).astype(np.uint32) if toc[-1] else np.zeros(0, np.uint32)
    return toc, kmer, synthetic.pack_markers(toc, kmer), kinds


def numpy_bound(toc, kmer, delta):
    """The screen's definition, slowly: pairs (i, j) of equal kmer ids with |i - j| < delta."""
    read_count = (len(toc) - 1) // 2
    out = np.zeros(read_count, np.uint32)
    for r in range(read_count):
        a = kmer[int(toc[2 * r]):int(toc[2 * r + 1])].astype(np.int64)
        b = kmer[int(toc[2 * r + 1]):int(toc[2 * r + 2])].astype(np.int64)
        n, count = len(a), 0
        for d in range(-(delta - 1), delta):
            lo, hi = max(0, -d), min(n, n - d)
            if hi > lo:
                count += int((a[lo:hi] == b[lo + d:hi + d]).sum())
        out[r] = count
    return out


class HostLib:
    """libshasta_mi355x_host.so (or its emulated twin): the host half, for unit parity of method 0."""

    def __init__(self, path):
        self.lib = C.CDLL(path)
        self.lib.shasta_mi355x_host_last_error.restype = C.c_char_p

    def self_alignment(self, k0, k1, max_skip=100, max_drift=100, max_marker_frequency=10):
        k0 = np.ascontiguousarray(k0, dtype=np.uint32)
        k1 = np.ascontiguousarray(k1, dtype=np.uint32)
        assert len(k0) == len(k1)
        cap = max(16, 16 * len(k0) + 16)
        out = np.zeros(2 * cap, dtype=np.uint32)
        count = C.c_uint64()
        rc = self.lib.shasta_mi355x_host_self_alignment_method0(
            k0.ctypes.data_as(C.POINTER(C.c_uint32)), k1.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(len(k0)),
            C.c_uint32(max_skip), C.c_uint32(max_drift), C.c_uint32(max_marker_frequency),
            out.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint64(cap), C.byref(count))
        if rc:
            raise RuntimeError(self.lib.shasta_mi355x_host_last_error().decode())
        return out[:2 * count.value].reshape(-1, 2)


def counts_of(alignment, delta_threshold):
    a = np.asarray(alignment, dtype=np.int64).reshape(-1, 2)
    return len(a), int((np.abs(a[:, 0] - a[:, 1]) < delta_threshold).sum())


PARAMETER_SETS = (
    DEFAULTS,
    dict(max_skip=30, max_drift=10, max_marker_frequency=3, aligned_fraction_threshold=0.05,
         near_diagonal_fraction_threshold=0.02, delta_threshold=20),
)
GOLDEN_READ_SET = dict(n_reads=64, seed=7)


def golden():
    """tests/golden/palindromic.npz: what the reference itself (AlignmentGraph.cpp compiled in place, oracle/_ref)
    answered for the real reads of tiny.npz and for read_set(**GOLDEN_READ_SET); made by make_golden_palindromic.py."""
    import hashlib
    import os
    from tests import support
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "palindromic.npz"))
    g = support.Golden("tiny.npz")
    toc, kmer, data7, kinds = read_set(**GOLDEN_READ_SET)
    assert hashlib.md5(kmer.tobytes()).hexdigest() == str(z["hairpins_input_md5"]), "the generator no longer reproduces the fixture's input"
    return z, {"tiny": (g.toc, g.data7), "hairpins": (toc, data7)}


def check_against_golden(flag_function):
    """flag_function(toc, data7, **parameters) -> (flags, aligned, near, digests) compared with the reference's answers."""
    z, inputs = golden()
    flagged = 0
    for name, (toc, data7) in inputs.items():
        for i, kw in enumerate(PARAMETER_SETS):
            got = flag_function(toc, data7, **kw)
            for what, g in zip(("flags", "aligned", "near", "digests"), got):
                if g is not None:
                    assert np.array_equal(np.asarray(g), z["%s_%d_%s" % (name, i, what)]), (name, i, what)
            flagged += int(z["%s_%d_flags" % (name, i)].sum())
    assert flagged > 20


def screen_is_sound(bound, toc, aligned_near, near_threshold):
    """The device bound never undercuts the reference's near-diagonal count, so screening on it is exact."""
    n = np.diff(np.asarray(toc, dtype=np.int64))[::2]
    assert np.all(bound.astype(np.int64) >= aligned_near.astype(np.int64))
    with np.errstate(invalid="ignore", divide="ignore"):
        screened = bound.astype(np.float64) / n.astype(np.float64) < near_threshold
    return screened


def flag_through_stage(toc, data7, tmp_path, host_library, initial_flags=None, **kw):
    """Assembler.flagPalindromicReads of the Python mirror on a Data/ directory -> (flags, counts)."""
    import os
    import shasta_amd.assembler as shasta
    from tests import host_support
    d = str(tmp_path / "Data")
    os.makedirs(d, exist_ok=True)
    shim = host_support.HostShim()
    read_count = (len(toc) - 1) // 2
    shim.write_data_dir(d, toc, data7, initial_flags if initial_flags is not None else np.zeros(read_count, np.uint8))
    a = shasta.Assembler(d, hostLibrary=host_library)
    a.accessMarkers()
    counts = a.flagPalindromicReads(kw["max_skip"], kw["max_drift"], kw["max_marker_frequency"], kw["aligned_fraction_threshold"],
                                    kw["near_diagonal_fraction_threshold"], kw["delta_threshold"], 2)
    stored, _ = shim.open_vector(os.path.join(d, "ReadFlags"), 1)
    return stored.reshape(-1), counts
