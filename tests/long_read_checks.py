"""Pairs of two reads beyond the largest LDS table class (8 192 markers) -- the ordinary pair of conf/Nanopore-UL-May2022.conf,
whose reads start at 50 000 bases -- through the windowed class of the cells stage (align4CellsLongKernel: the shorter read tabled
in windows of 2^13 markers, the matches listed) and what follows it: the sort kernel's classes for reads of up to 32 768 markers,
the wave kernel's large classes, the anchor kernel.  Checked against the oracle (and the reference's own code where it is built);
the kernel table says which kernels the candidates took.  Shared by the -m gpu tests and their pre-flight on the emulated build."""
import os

import numpy as np

from shasta_amd import abi, synthetic
from tests import adversarial, support

UL_ALIGN = dict(maxSkip=100, maxDrift=100, maxTrim=100, minAlignedMarkerCount=10, minAlignedFraction=0.1)


def noisy(rng, x, keep=0.8, spurious=0.2, alphabet=None):
    x = x[rng.random(len(x)) < keep]
    extra = rng.random(len(x)) < spurious
    idx = np.repeat(np.arange(len(x)), 1 + extra.astype(np.int64))
    copy = np.zeros(len(idx), dtype=bool)
    copy[1:] = idx[1:] == idx[:-1]
    y = x[idx].copy()
    y[copy] = alphabet[rng.integers(0, len(alphabet), size=int(copy.sum()))] if alphabet is not None else rng.integers(0, 1 << 28, size=int(copy.sum()), dtype=np.uint32)
    return y.astype(np.uint32)


def long_read_set(seed=66, lengths=(9000, 12500, 17000, 9500, 24000, 8300, 33500, 10100, 4000, 15000), genome_markers=42000, alphabet_size=None):
    """Reads cut from one genome at overlapping places; alphabet_size = None: k = 14-like ids (hardly any random match);
    a number: that many distinct ids (a background of random matches as at k = 10)."""
    rng = np.random.default_rng(seed)
    alphabet = None if alphabet_size is None else rng.choice(1 << 20, size=alphabet_size, replace=False).astype(np.uint32)
    genome = rng.integers(0, 1 << 28, size=genome_markers, dtype=np.uint32) if alphabet is None else alphabet[rng.integers(0, alphabet_size, size=genome_markers)]
    reads = []
    for i, n in enumerate(lengths):
        span = int(n / 0.96)                                   # (0.8 kept, 0.2 doubled: 0.96 markers out per marker in)
        start = int(rng.integers(0, max(1, genome_markers - span)))
        if i % 3 == 0:
            start = min(start, 3000)                           # several reads over the genome's beginning: long overlaps
        reads.append(noisy(rng, genome[start:start + span], alphabet=alphabet))
    return adversarial.build(reads)


def kernel_rows(ctx):
    return {name: r for name, r in ctx.kernel_table().items() if r["launches"]}


def both_long(lib, oracle_lib, ref_lib=None, alphabet_size=None, seed=66, **kw):
    """Every pair of the set's reads on both strands: LDS windowed class for the pairs of two long reads, parity with the oracle."""
    toc, kmer, data7 = long_read_set(seed=seed, alphabet_size=alphabet_size, **kw)
    n_reads = (len(toc) - 1) // 2
    cand = adversarial.all_pairs(n_reads)
    o = abi.default_align4_options(**UL_ALIGN)
    x = oracle_lib.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
    with lib.context(0) as ctx:
        ctx.set_markers(toc, data7)
        ctx.kernel_table_reset()
        y = ctx.align4(cand, o, want_ordinals=True)
        rows = kernel_rows(ctx)
    ties = (x.status & 0x80) != 0
    assert x.per_candidate(~ties) == y.per_candidate(~ties)
    if ref_lib is not None:
        support.same_align(ref_lib.align4_batch(toc, data7, cand, o, want_ordinals=True), y)
    sizes = np.diff(toc.astype(np.int64))[0::2]
    n0, n1 = sizes[cand["readId0"]], sizes[cand["readId1"]]
    expected_long = int(((n0 > 8191) & (n1 > 8191) & (n0 < 65535) & (n1 < 65535)).sum())
    long_rows = [r for name, r in rows.items() if name.startswith("align4CellsLongKernel")]
    hbm_rows = [r for name, r in rows.items() if name.startswith("align4CellsKernel")]
    return {"candidates": len(cand), "stored": int((x.status == abi.SHASTA_ALIGN_STORED).sum()), "both_long": expected_long,
            "in_the_windowed_class": int(sum(r["work"] for r in long_rows)), "in_the_hbm_scratch_kernel": int(sum(r["work"] for r in hbm_rows)),
            "dense_because": {name: int(r["launches"]) for name, r in rows.items() if name.startswith("dense DP because")}, "rows": rows}


def full_tables(lib, oracle_lib, ref_lib=None, lengths=(8600, 9100), alphabet_size=400, seed=71):
    """Repeat-rich pairs of two long reads (a few hundred distinct marker ids: every cell of the alignment matrix collects matches, as
    between real reads at k = 10): the windowed class's cell table fills, the candidate climbs to the kernel with its tables in device
    memory, whose first tables fill as well.  A full table ends the candidate's counting there and then (round 6: every further match
    walked the whole table before saying so again -- 36 s per candidate on the emulated build, `slots` compare-and-swaps in device
    memory per match in the HBM-scratch kernel); what is checked is that the climb still ends in the reference's answer."""
    toc, kmer, data7 = long_read_set(seed=seed, lengths=lengths, genome_markers=int(1.3 * max(lengths)), alphabet_size=alphabet_size)
    n_reads = (len(toc) - 1) // 2
    cand = adversarial.all_pairs(n_reads)
    o = abi.default_align4_options(**UL_ALIGN)
    x = oracle_lib.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
    with lib.context(0) as ctx:
        ctx.set_markers(toc, data7)
        ctx.kernel_table_reset()
        import time
        t0 = time.perf_counter()
        y = ctx.align4(cand, o, want_ordinals=True)
        seconds = time.perf_counter() - t0
        rows = kernel_rows(ctx)
    ties = (x.status & 0x80) != 0
    assert x.per_candidate(~ties) == y.per_candidate(~ties)
    if ref_lib is not None:
        support.same_align(ref_lib.align4_batch(toc, data7, cand, o, want_ordinals=True), y)
    return {"candidates": len(cand), "aligner_seconds": seconds,
            "windowed_launches": int(sum(r["launches"] for name, r in rows.items() if name.startswith("align4CellsLongKernel"))),
            "hbm_scratch_launches": int(sum(r["launches"] for name, r in rows.items() if name.startswith("align4CellsKernel"))),
            "hbm_scratch_candidates": int(sum(r["work"] for name, r in rows.items() if name.startswith("align4CellsKernel")))}


def forced(lib, oracle_lib, ref_lib, force, n_reads=200, limit=1500, adversarial_sets=True):
    """SHASTA_MI355X_CELLS_FORCE=long / big: every candidate the windowed class can take starts in it (or in its large-graph form) --
    ordinary reads and the adversarial read sets through align4CellsLongKernel / align4CellsLongBigKernel, against the reference's own
    Align4 (component ties included: the reference resolves them, the library must pick the same component)."""
    os.environ["SHASTA_MI355X_CELLS_FORCE"] = force
    try:
        checked = 0
        toc, kmer, data7 = support.small_marker_set(n_reads=n_reads, genome_markers=12000, seed=99)
        p = abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30)
        cand = oracle_lib.lowhash0(toc, data7, None, p).candidates[:limit]
        for o in (abi.default_align4_options(minAlignedMarkerCount=40), abi.default_align4_options(**UL_ALIGN)):
            with lib.context(0) as ctx:
                ctx.set_markers(toc, data7)
                ctx.kernel_table_reset()
                x = ctx.align4(cand, o, want_ordinals=True)
                rows = kernel_rows(ctx)
            name = "align4CellsLongBigKernel" if force == "big" else "align4CellsLongKernel<false>"
            assert name in rows and rows[name]["work"] >= len(cand) // 2, rows.keys()
            y = (ref_lib or oracle_lib).align4_batch(toc, data7, cand, o, want_ordinals=True)
            if ref_lib is not None:
                support.same_align(x, y)
            else:
                ties = (y.status & 0x80) != 0
                assert x.per_candidate(~ties) == y.per_candidate(~ties)
            checked += len(cand)
        if adversarial_sets:
            for _, reads in adversarial.read_sets(long_reads=False):
                toc, kmer, data7 = adversarial.build(reads)
                cand = adversarial.all_pairs(len(reads))
                o = abi.default_align4_options(minAlignedMarkerCount=10)
                x = lib.align4_batch(toc, data7, cand, o, want_ordinals=True)
                y = (ref_lib or oracle_lib).align4_batch(toc, data7, cand, o, want_ordinals=True)
                ties = (y.status & 0x80) != 0
                assert x.per_candidate(~ties) == y.per_candidate(~ties) and np.array_equal(x.status[~ties], y.status[~ties])
                checked += len(cand)
        return checked
    finally:
        del os.environ["SHASTA_MI355X_CELLS_FORCE"]
