"""Parity checks of align method 3 shared by the GPU tests (tests/test_gpu_align3_and_markers.py) and
their pre-flight on the emulated build (tests/test_emu_kernels.py): `lib` is a
shasta_amd.lib.Library, everything is compared bit for bit."""
import numpy as np

from shasta_amd import abi
from tests import support


def golden_fixture(lib, name, i):
    """Fixtures made by the reference's own alignOrientedReads3 (tests/golden/make_golden_align3.py)."""
    g = support.Golden(name + ".npz")
    z = np.load(support.GOLDEN + "/" + name + "_align3.npz")
    o = abi.default_align3_options(**support.ALIGN3_OPTION_SETS[i])
    out = lib.align3_batch(g.toc, g.data7, g.candidates(0), o, want_ordinals=True)
    support.check_align3(out, z, i)
    return out


def against_oracle(lib, oracle_lib, seed, kw, n_reads=150, genome_markers=12000, limit=300):
    toc, kmer, data7 = support.small_marker_set(n_reads=n_reads, genome_markers=genome_markers, seed=seed)
    p = abi.default_lowhash0_params(minBucketSize=3, maxBucketSize=30, minFrequency=2)
    cand = oracle_lib.lowhash0(toc, data7, None, p).candidates[:limit]
    assert len(cand) > 50
    o = abi.default_align3_options(**kw)
    ref = oracle_lib.align3_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
    out = lib.align3_batch(toc, data7, cand, o, want_ordinals=True)
    support.same_align(ref, out)
    assert np.array_equal(ref.compressed_data, out.compressed_data)
    return ref, out


def context_paths(lib, oracle_lib, seed=31):
    """Resident markers: method 3 after LowHash0 on one context, owned and borrowed results, the
    down-sampled markers rebuilt when k or the factor change, method 4 unaffected in between."""
    toc, kmer, data7 = support.small_marker_set(n_reads=120, genome_markers=9000, seed=seed)
    p = abi.default_lowhash0_params(minBucketSize=3, maxBucketSize=30, minFrequency=2)
    o3 = abi.default_align3_options(minAlignedMarkerCount=40)
    o3b = abi.default_align3_options(minAlignedMarkerCount=40, downsamplingFactor=0.2, k=8)
    o4 = abi.default_align4_options(minAlignedMarkerCount=40)
    with lib.context(0) as ctx:
        ctx.set_kmer_ids(toc, kmer)
        cand = ctx.lowhash0(p).candidates[:200]
        a = ctx.align3(cand, o3, want_ordinals=True)
        m4 = ctx.align4(cand, o4, want_ordinals=True)
        b = ctx.align3(cand, o3b, want_ordinals=True)
        c = ctx.align3(cand, o3, want_ordinals=False, borrow=True)
        c_status, c_rows, c_bytes = c.status.copy(), c.info_table(), c.compressed_data.copy()
        del c
    for out, o in ((a, o3), (b, o3b)):
        ref = oracle_lib.align3_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
        support.same_align(ref, out)
    ref4 = oracle_lib.align4_batch(toc, data7, cand, o4, want_ordinals=True, threads=0)
    if not (ref4.status & 0x80).any():
        support.same_align(ref4, m4)
    assert np.array_equal(c_status, a.status) and np.array_equal(c_rows, a.info_table())
    assert np.array_equal(c_bytes, a.compressed_data)


def rejected_options(lib):
    """What this version does not do fails loudly instead of computing something else."""
    import pytest
    toc, kmer, data7 = support.small_marker_set(n_reads=20, genome_markers=3000, seed=1)
    cand = abi.make_pairs([0], [1], [1])
    for kw in (dict(gapScore=-(1 << 21)), dict(matchScore=1 << 25), dict(maxBand=70000), dict(k=17), dict(downsamplingFactor=1.5),
               dict(bandExtend=-1)):
        with pytest.raises(RuntimeError, match="Align3"):
            lib.align3_batch(toc, data7, cand, abi.default_align3_options(**kw))
    with pytest.raises(RuntimeError, match="invalid alignment candidate"):
        lib.align3_batch(toc, data7, abi.make_pairs([3], [3], [1]), abi.default_align3_options())
    out = lib.align3_batch(toc, data7, abi.make_pairs([], [], []), abi.default_align3_options())
    assert len(out.alignment_data) == 0 and len(out.status) == 0


def long_reads(lib, oracle_lib, seed=41, mean_markers=18000.0, factors=(0.5, 0.1), cases=(0, 1)):
    """Step 1 of pairs whose down-sampled matrix has more diagonals than a register-resident DP task holds (1024: the
    LDS-row kernel) and more than its LDS rows hold (8192: rows in HBM scratch) -- the reference has no limit there."""
    from shasta_amd import synthetic
    toc, kmer = synthetic.marker_reads(8, int(1.4 * mean_markers), mean_markers=mean_markers, sigma=0.05, min_markers=int(0.85 * mean_markers), seed=seed)
    data7 = synthetic.pack_markers(toc, kmer)
    lengths = np.diff(toc.astype(np.int64))[0::2]
    order = np.argsort(-lengths)
    a, b, c = (int(x) for x in order[:3])
    cand = abi.make_pairs([min(a, b), min(a, c), min(b, c)], [max(a, b), max(a, c), max(b, c)], [1, 1, 0])
    for kw, least in [((dict(downsamplingFactor=factors[0], minAlignedMarkerCount=40), 8192), (dict(downsamplingFactor=factors[1], minAlignedMarkerCount=40), 1024))[c] for c in cases]:
        o = abi.default_align3_options(**kw)
        ref = oracle_lib.align3_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
        out = lib.align3_batch(toc, data7, cand, o, want_ordinals=True)
        support.same_align(ref, out)
        assert np.array_equal(ref.compressed_data, out.compressed_data)
        assert ((out.status & 0x7f) != abi.SHASTA_ALIGN_SKIPPED).all()
        assert float(kw["downsamplingFactor"]) * (lengths[b] + lengths[c]) > 1.1 * least       # the path the case is meant for
    return out
