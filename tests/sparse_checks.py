"""The sparse form of the banded alignment (shasta_amd/csrc/align4_sparse.hpp: the alignment from the matches inside the band,
answered only where the optimal chain of matches is unique) against the dense kernels and the oracle: the same results whether
the sparse path is on or off and under every compiled tie policy, most tasks of clean reads certified, tie-heavy and repeat-rich
tasks handed to the dense kernels; and, where several optimal chains tie only locally, the anchor kernel (align4_anchor.hpp: the
dense DP on the rectangles between the matches every optimal alignment holds).  Shared by the -m gpu tests and their pre-flight on the
emulated build."""
import os

import numpy as np

from shasta_amd import abi
from tests import dp_geometry_checks, support, tie_policy_checks


class switched_off:
    def __enter__(self):
        self.previous = os.environ.get("SHASTA_MI355X_SPARSE_DP")
        os.environ["SHASTA_MI355X_SPARSE_DP"] = "0"

    def __exit__(self, *exc):
        if self.previous is None:
            del os.environ["SHASTA_MI355X_SPARSE_DP"]
        else:
            os.environ["SHASTA_MI355X_SPARSE_DP"] = self.previous


def clean_tasks(seed, tasks=90, long_every=11):
    """Noisy copies over a large alphabet (a unique optimal chain nearly always), bands of every class, a few contained and
    barely overlapping pairs, stream reads on either side of the sparse path's limit of 8192 markers."""
    rng = np.random.default_rng(seed)
    pieces, spec, at = [], [], 0
    for t in range(tasks):
        width = int(rng.choice([10, 20, 40, 50, 60, 80, 100, 200, 500, 1000]))
        n = int(rng.integers(50, 1500)) if t % long_every else int(rng.integers(8000, 9000))
        genome = rng.integers(0, 1 << 20, size=2 * n + 400, dtype=np.uint32)
        off = int(rng.integers(0, 200))
        a = dp_geometry_checks.noisy(rng, genome[:n], 1 << 20)
        b = dp_geometry_checks.noisy(rng, genome[off:off + int(rng.integers(n // 3, n + 1))], 1 << 20)
        if t % 7 == 0:
            a, b = b, a
            off = -off
        lo = off - width // 2 + int(rng.integers(-8, 8))
        lo = min(max(lo, -len(b) - width + 1), len(a))
        pieces += [a, b]
        spec.append((at, len(a), at + len(a), len(b), lo, lo + width - 1))
        at += len(a) + len(b)
    return np.concatenate(pieces), np.asarray(spec, dtype=np.int64)


def _run(lib, kmer, spec, timing=False):
    return lib.banded_dp_many(kmer, spec[:, 0], spec[:, 1], spec[:, 2], spec[:, 3], spec[:, 4], spec[:, 5], timing=timing)


def dp_tasks(lib, orc, seed=71, clean=90, tie_heavy=60, alternatives=tie_policy_checks.ALTERNATIVES, long_every=11):
    """-> (tasks, share of the DP cells that never reached the dense kernels on clean tasks, the same on tie-heavy ones)."""
    shares = []
    for kmer, spec in (clean_tasks(seed, tasks=clean, long_every=long_every), tie_policy_checks.tie_heavy_tasks(seed + 1, tasks=tie_heavy)):
        want = [orc.banded_dp(kmer[b0:b0 + nx], kmer[b1:b1 + ny], int(lo), int(hi)) for b0, nx, b1, ny, lo, hi in spec]
        got = _run(lib, kmer, spec)
        with switched_off():
            dense = _run(lib, kmer, spec)
            cells_dense = int(_run(lib, kmer, spec, timing=True)[3].sum())
        cells_left = int(_run(lib, kmer, spec, timing=True)[3].sum())
        for (x, sx), (y, sy), (z, sz) in zip(want, got, dense):
            assert sx == sy == sz and np.array_equal(x, y) and np.array_equal(x, z)
        # Narrow tasks only are booked per class; all of them when the sparse path is off.
        narrow = (spec[:, 5] - spec[:, 4] + 1) <= 1024
        assert cells_dense == int((spec[narrow, 1] * (spec[narrow, 5] - spec[narrow, 4] + 1)).sum())
        shares.append(1.0 - cells_left / max(1, cells_dense))
        # Under the other compiled tie policies: certified tasks do not depend on the policy, the others follow it.
        for alternative in alternatives:
            with tie_policy_checks.policy(orc, alternative):
                want_alt = [orc.banded_dp(kmer[b0:b0 + nx], kmer[b1:b1 + ny], int(lo), int(hi)) for b0, nx, b1, ny, lo, hi in spec]
                got_alt = _run(lib, kmer, spec)
            for (x, sx), (y, sy) in zip(want_alt, got_alt):
                assert sx == sy and np.array_equal(x, y)
    return len(spec), shares[0], shares[1]


def aligner(lib, orc, n_reads=160, limit=500):
    """Align4 on candidates of synthetic reads, the sparse path on and off: equal to each other and to the oracle; with it on, most
    DP cells never reach the forward kernel."""
    toc, kmer, data7 = support.small_marker_set(n_reads=n_reads, genome_markers=12000, seed=73)
    cand = orc.lowhash0(toc, data7, None, abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30, minFrequency=1)).candidates[:limit]
    o = abi.default_align4_options(minAlignedMarkerCount=40)
    want = orc.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
    rows = {}
    with lib.context(0) as ctx:
        ctx.set_kmer_ids(toc, kmer)
        for name in ("sparse", "sparse, streaks from the pairs", "dense"):
            ctx.kernel_table_reset()
            if name == "dense":
                with switched_off():
                    got = ctx.align4(cand, o, want_ordinals=True)
            elif name == "sparse":
                got = ctx.align4(cand, o, want_ordinals=True)
            else:
                # (the form before: compressWriteKernel makes the streaks of the wave kernel's alignments from their aligned pairs, not
                # by copying what the wave kernel wrote as it walked the chain)
                with _environment(SHASTA_MI355X_CHAIN_WAVE_STREAM="0"):
                    got = ctx.align4(cand, o, want_ordinals=True)
            table = ctx.kernel_table()
            rows[name] = sum(v["work"] for k, v in table.items() if k.startswith("bandedDpForwardKernel"))
            assert ("sparseChainWaveKernel" in table or "sparseChainKernel" in table) == name.startswith("sparse")
            ties = (want.status & 0x80) != 0
            if not ties.any():
                support.same_align(want, got)
            else:
                assert want.per_candidate(~ties) == got.per_candidate(~ties)
            assert got.dp_cell_count == want.dp_cell_count
    assert rows["dense"] == want.dp_cell_count
    return 1.0 - rows["sparse"] / max(1, rows["dense"])


class anchors_off:
    def __enter__(self):
        self.previous = os.environ.get("SHASTA_MI355X_ANCHORED_DP")
        os.environ["SHASTA_MI355X_ANCHORED_DP"] = "0"

    def __exit__(self, *exc):
        if self.previous is None:
            del os.environ["SHASTA_MI355X_ANCHORED_DP"]
        else:
            os.environ["SHASTA_MI355X_ANCHORED_DP"] = self.previous


def locally_ambiguous_tasks(seed, tasks=30):
    """Noisy copies over a large alphabet with markers doubled in place and short stretches copied right behind themselves (what a
    homopolymer run or a tandem repeat leaves among the markers): the optimal chain is unique except around those places, where two
    or more sub-chains tie -- at the very ends of the reads too."""
    rng = np.random.default_rng(seed)
    pieces, spec, at = [], [], 0

    def doubled(s, frac):
        s = np.repeat(s, 1 + (rng.random(len(s)) < frac).astype(np.int64) + (rng.random(len(s)) < frac / 4).astype(np.int64))
        for _ in range(int(rng.integers(0, 4))):
            if len(s) < 20:
                break
            p, length = int(rng.integers(0, len(s) - 8)), int(rng.integers(2, 7))
            s = np.concatenate([s[:p + length], s[p:p + length], s[p + length:]])
        return s

    for t in range(tasks):
        width = int(rng.choice([12, 20, 40, 60, 100, 200, 400]))
        n = int(rng.integers(30, 1200))
        alphabet = int(rng.choice([1 << 20, 1 << 20, 1 << 20, 5000]))
        genome = rng.integers(0, alphabet, size=2 * n + 400, dtype=np.uint32)
        off = int(rng.integers(0, 200))
        frac = float(rng.choice([0.003, 0.01, 0.03]))
        a = doubled(dp_geometry_checks.noisy(rng, genome[:n], alphabet), frac)
        b = doubled(dp_geometry_checks.noisy(rng, genome[off:off + int(rng.integers(n // 3, n + 1))], alphabet), frac)
        if t % 6 == 1:                                   # an ambiguity at the first and at the last marker of both reads
            a = np.concatenate([a[:1], a, a[-1:]])
        if t % 6 == 2:
            b = np.concatenate([b[:1], b[:1], b, b[-1:]])
        if t % 5 == 0:
            a, b = b, a
            off = -off
        lo = off - width // 2 + int(rng.integers(-8, 8))
        lo = min(max(lo, -len(b) - width + 1), len(a))
        pieces += [a, b]
        spec.append((at, len(a), at + len(a), len(b), lo, lo + width - 1))
        at += len(a) + len(b)
    return np.concatenate(pieces), np.asarray(spec, dtype=np.int64)


def anchored_tasks(lib, orc, seeds=(3, 4), tasks=30, alternatives=tie_policy_checks.ALTERNATIVES):
    """-> (task runs, DP cells of the tasks, cells the dense kernels ran with the sparse path alone, cells with the anchor kernel too).
    Equal to the oracle under the default and every alternative tie policy (the rectangles follow the policy), and with the anchor
    kernel switched off."""
    runs, cells_all, cells_sparse, cells_anchored = 0, 0, 0, 0
    for seed in seeds:
        kmer, spec = locally_ambiguous_tasks(seed, tasks=tasks)
        for policy in (None,) + tuple(alternatives):
            if policy is None:
                want = [orc.banded_dp(kmer[b0:b0 + nx], kmer[b1:b1 + ny], int(lo), int(hi)) for b0, nx, b1, ny, lo, hi in spec]
                got = _run(lib, kmer, spec)
                with anchors_off():
                    without = _run(lib, kmer, spec)
                    cells_sparse += int(_run(lib, kmer, spec, timing=True)[3].sum())
                cells_anchored += int(_run(lib, kmer, spec, timing=True)[3].sum())
                for (x, sx), (z, sz) in zip(want, without):
                    assert sx == sz and np.array_equal(x, z)
            else:
                with tie_policy_checks.policy(orc, policy):
                    want = [orc.banded_dp(kmer[b0:b0 + nx], kmer[b1:b1 + ny], int(lo), int(hi)) for b0, nx, b1, ny, lo, hi in spec]
                    got = _run(lib, kmer, spec)
            for i, ((x, sx), (y, sy)) in enumerate(zip(want, got)):
                assert sx == sy and np.array_equal(x, y), (seed, policy, tuple(int(v) for v in spec[i]))
            runs += len(spec)
        narrow = (spec[:, 5] - spec[:, 4] + 1) <= 1024
        cells_all += int((spec[narrow, 1] * (spec[narrow, 5] - spec[narrow, 4] + 1)).sum())
    return runs, cells_all, cells_sparse, cells_anchored


def tiny_tasks(lib, orc, seed=11, tasks=400, alternatives=tie_policy_checks.ALTERNATIVES):
    """Reads of 1 to 40 markers over alphabets of 2^20, 6 and 3 k-mers, bands of 4 to 64 diagonals anywhere in the matrix: tasks
    with no hit, one hit, fewer hits than the chain kernel takes in its first group, and ties everywhere.  -> task runs."""
    rng = np.random.default_rng(seed)
    pieces, spec, at = [], [], 0
    for t in range(tasks):
        n = int(rng.integers(1, 40))
        alphabet = int(rng.choice([1 << 20, 6, 3]))
        g = rng.integers(0, alphabet, size=n + 10, dtype=np.uint32)
        a = g[:n].copy()
        b = g[int(rng.integers(0, 5)):][:int(rng.integers(1, n + 1))].copy()
        if rng.random() < 0.3:
            b = np.repeat(b, 1 + (rng.random(len(b)) < 0.2))
        w = int(rng.choice([4, 8, 20, 64]))
        lo = int(rng.integers(-len(b), len(a))) - w // 2
        lo = min(max(lo, -len(b) - w + 1), len(a))
        pieces += [a, b]
        spec.append((at, len(a), at + len(a), len(b), lo, lo + w - 1))
        at += len(a) + len(b)
    kmer, spec = np.concatenate(pieces), np.asarray(spec, dtype=np.int64)
    runs = 0
    for policy in (0,) + tuple(alternatives):
        with tie_policy_checks.policy(orc, policy):
            want = [orc.banded_dp(kmer[b0:b0 + nx], kmer[b1:b1 + ny], int(lo), int(hi)) for b0, nx, b1, ny, lo, hi in spec]
            got = _run(lib, kmer, spec)
        for i, ((x, sx), (y, sy)) in enumerate(zip(want, got)):
            assert sx == sy and np.array_equal(x, y), (policy, tuple(int(v) for v in spec[i]))
        runs += len(spec)
    return runs


class _environment:
    def __init__(self, **values):
        self.values = values

    def __enter__(self):
        self.previous = {k: os.environ.get(k) for k in self.values}
        os.environ.update(self.values)

    def __exit__(self, *exc):
        for k, v in self.previous.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def large_tasks(seed=5, tasks=16):
    """Tasks of 1 100 to 3 100 aligned pairs: beyond the wave kernel's first capacity class (1 024 hits in LDS), some beyond its
    second (2 048), lists of more than 1 024 matches (the sort kernel's passes over memory)."""
    rng = np.random.default_rng(seed)
    pieces, spec, at = [], [], 0
    for t in range(tasks):
        n = int(rng.integers(2000, 4600))
        width = int(rng.choice([40, 60, 100, 300]))
        genome = rng.integers(0, 1 << 20, size=2 * n + 400, dtype=np.uint32)
        off = int(rng.integers(0, 100))
        a = dp_geometry_checks.noisy(rng, genome[:n], 1 << 20)
        b = dp_geometry_checks.noisy(rng, genome[off:off + n], 1 << 20)
        lo = off - width // 2
        pieces += [a, b]
        spec.append((at, len(a), at + len(a), len(b), lo, lo + width - 1))
        at += len(a) + len(b)
    return np.concatenate(pieces), np.asarray(spec, dtype=np.int64)


def huge_tasks(seed=6, sizes=(7000, 9000, 11500, 14000)):
    """Tasks of 4 300 to 8 700 aligned pairs between two reads of up to 10 000 markers: the wave kernel's classes of 5 456 hits (D in 16
    bits) and of 8 000 and 15 360 (32), hits ordered by a read beyond 8 192 markers (the sort kernel's classes for long reads)."""
    rng = np.random.default_rng(seed)
    pieces, spec, at = [], [], 0
    for n in sizes:
        width = int(rng.choice([60, 100, 300]))
        genome = rng.integers(0, 1 << 20, size=n + 400, dtype=np.uint32)
        off = int(rng.integers(0, 100))
        a = dp_geometry_checks.noisy(rng, genome[:n], 1 << 20)
        b = dp_geometry_checks.noisy(rng, genome[off:off + n], 1 << 20)
        lo = off - width // 2
        pieces += [a, b]
        spec.append((at, len(a), at + len(a), len(b), lo, lo + width - 1))
        at += len(a) + len(b)
    return np.concatenate(pieces), np.asarray(spec, dtype=np.int64)


def deep_block_tasks(seed=23, depths=(2500, 5200, 7700), n=8100, block=320):
    """Two unrelated reads of 8 100 markers (the longest the hits are ordered by: 8 192) that share one block deep inside: the chain
    enters from the border at a cost of the block's depth, so D of its hits lies near -depth -- the low end of what the wave kernel
    keeps in 16 bits."""
    rng = np.random.default_rng(seed)
    pieces, spec, at = [], [], 0
    for depth in depths:
        a = rng.integers(0, 1 << 20, size=n, dtype=np.uint32)
        b = rng.integers(0, 1 << 20, size=n, dtype=np.uint32)
        shift = int(rng.integers(-20, 20))
        copy = dp_geometry_checks.noisy(rng, a[depth:depth + block], 1 << 20)
        b[depth + shift:depth + shift + len(copy)] = copy
        pieces += [a, b]
        spec.append((at, n, at + n, n, -shift - 40, -shift + 40))
        at += 2 * n
    return np.concatenate(pieces), np.asarray(spec, dtype=np.int64)


def wave_kernel_forms(lib, orc):
    """align4_chainwave.hpp against the oracle and against the forms it can be switched to: the lane-per-task chain kernel
    (SHASTA_MI355X_CHAIN_WAVE=0), the wave kernel ordering the hits itself (SHASTA_MI355X_CHAIN_WAVE_SORT=1: classes chosen from the
    listed matches, tasks that turn out too large handed to the next class), D in 32 bits (SHASTA_MI355X_CHAIN_WAVE_WIDE_D=1), on clean
    tasks of every size class and on blocks deep inside long reads.
    -> tasks compared."""
    compared = 0
    for kmer, spec in (clean_tasks(91, tasks=40, long_every=9), large_tasks(), huge_tasks(), deep_block_tasks()):
        want = [orc.banded_dp(kmer[b0:b0 + nx], kmer[b1:b1 + ny], int(lo), int(hi)) for b0, nx, b1, ny, lo, hi in spec]
        for env in ({}, {"SHASTA_MI355X_CHAIN_WAVE": "0"}, {"SHASTA_MI355X_CHAIN_WAVE_SORT": "1"}, {"SHASTA_MI355X_CHAIN_WAVE_SIDE": "1"},
                    {"SHASTA_MI355X_CHAIN_WAVE_WIDE_D": "1"}, {"SHASTA_MI355X_CHAIN_WAVE_WIDE_D": "1", "SHASTA_MI355X_CHAIN_WAVE_SORT": "1"}):
            with _environment(**env):
                got = _run(lib, kmer, spec)
            for (x, sx), (y, sy) in zip(want, got):
                assert sx == sy and np.array_equal(x, y), env
            compared += len(want)
    return compared


def inverted_block_tasks(blocks=(20, 70, 71, 120, 300, 3200, 4000, 6000), flank=220, seed=17):
    """Two reads that agree except for one block of m markers which the second read has in REVERSED order: every chain can take one
    match of the block, the two in its middle tie (m even) -- two optimal chains, and between the anchors before and behind the block
    a rectangle of (m + 1)^2 cells: within the anchor kernel's first launch (4 096 cells) for m = 20, its second (the band's cells at
    two bits each -- 393 216 of them; sides of any length since round 6: 3 071 markers at most before) for 70 to 4 000, beyond both
    for 6 000 (486 000 cells of the band: the dense kernels)."""
    rng = np.random.default_rng(seed)
    pieces, spec, at = [], [], 0
    for m in blocks:
        ids = rng.permutation(1 << 20)[:2 * flank + m].astype(np.uint32)
        a = ids
        b = np.concatenate([ids[:flank], ids[flank:flank + m][::-1], ids[flank + m:]])
        pieces += [a, b]
        spec.append((at, len(a), at + len(a), len(b), -40, 40))
        at += len(a) + len(b)
    # Rectangles that end at the FREE border, too large for the first launch: the second read stops (or begins) inside the first one
    # with its last (first) matching marker doubled -- two optimal ends -- and a few markers of its own beyond, so that the window
    # between the last anchor and the border is 600 markers of the first read by 20 of the second, nearly all of it outside the band.
    # (... of 600 markers, and -- round 6: sides beyond 3 071 markers -- of 4 600 and 4 400: the free border on either side of a long read.)
    for where, total in (("end", 1300), ("begin", 1300), ("end", 5300), ("begin", 5000)):
        g = rng.permutation(1 << 20)[:total].astype(np.uint32)
        own = rng.permutation(1 << 20)[:20].astype(np.uint32) + np.uint32(1 << 21)          # (ids the other read does not hold)
        a = g
        if where == "end":
            b = np.concatenate([g[400:700], g[699:700], own])
            diagonal = 400
        else:
            first = total - 700                                                               # (the second read begins this far into the first)
            b = np.concatenate([own, g[first:first + 1], g[first:first + 300]])
            diagonal = first - 21
        pieces += [a, b]
        spec.append((at, len(a), at + len(a), len(b), diagonal - 30, diagonal + 30))
        at += len(a) + len(b)
    return np.concatenate(pieces), np.asarray(spec, dtype=np.int64)


def anchor_kernel_second_launch(lib, orc, alternatives=tie_policy_checks.ALTERNATIVES):
    """-> (DP cells the dense kernels ran with the anchor kernel's second launch, without it).  Equal to the oracle either way and
    under every compiled tie policy (the rectangle's walk follows the policy)."""
    kmer, spec = inverted_block_tasks()
    cells = {}

    def compare(want, record):
        for big in ("1", "0"):
            with _environment(SHASTA_MI355X_ANCHOR_BIG=big):
                got = _run(lib, kmer, spec)
                if record:
                    cells[big] = int(_run(lib, kmer, spec, timing=True)[3].sum())
            for (x, sx), (y, sy) in zip(want, got):
                assert sx == sy and np.array_equal(x, y), big

    compare([orc.banded_dp(kmer[b0:b0 + nx], kmer[b1:b1 + ny], int(lo), int(hi)) for b0, nx, b1, ny, lo, hi in spec], True)
    for alternative in alternatives:
        with tie_policy_checks.policy(orc, alternative):
            compare([orc.banded_dp(kmer[b0:b0 + nx], kmer[b1:b1 + ny], int(lo), int(hi)) for b0, nx, b1, ny, lo, hi in spec], False)
    return cells["1"], cells["0"]


def without_ordinals(lib, oracle_lib, n_reads=200, limit=1200):
    """A call that does not ask for the ordinals (Assembler::computeAlignments stores AlignmentData and the compressed alignments only):
    the wave kernel then does not write the aligned pairs of its tasks at all -- rows and compressed bytes must be what they are with them."""
    toc, kmer, data7 = support.small_marker_set(n_reads=n_reads, genome_markers=12000, seed=31)
    cand = oracle_lib.lowhash0(toc, data7, None, abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30)).candidates[:limit]
    o = abi.default_align4_options(minAlignedMarkerCount=40)
    x = oracle_lib.align4_batch(toc, data7, cand, o, want_ordinals=True, threads=0)
    with lib.context(0) as ctx:
        ctx.set_markers(toc, data7)
        y = ctx.align4(cand, o, want_ordinals=False)
        z = ctx.align4(cand, o, want_ordinals=True)
    for r in (y, z):
        assert np.array_equal(r.status & 0x7f, x.status & 0x7f) and np.array_equal(r.info_table(), x.info_table())
        assert np.array_equal(r.compressed_toc, x.compressed_toc) and np.array_equal(r.compressed_data, x.compressed_data)
    assert np.array_equal(z.ordinals, x.ordinals)
    return int((x.status == abi.SHASTA_ALIGN_STORED).sum())


def long_dense_paths(lib, orc, seed=91):
    """Tasks of 2 048 iterations and more with the sparse path OFF: the dense forward kernel of every band class, then the walk of a
    wavefront per task (dpTracebackWaveKernel: the trace through LDS in blocks of 32 chunks) -- paths that end at either border, small
    alphabets (ties), bundles of several such tasks."""
    rng = np.random.default_rng(seed)
    pieces, spec, at = [], [], 0
    for width in (20, 40, 64, 80, 128, 250, 500, 1000, 30, 300):
        for alphabet in ((1 << 20), 12):
            n = int(rng.integers(3600, 4400))
            genome = rng.integers(0, alphabet, size=2 * n + 600, dtype=np.uint32)
            off = int(rng.integers(0, 400))
            a = dp_geometry_checks.noisy(rng, genome[:n], alphabet)
            b = dp_geometry_checks.noisy(rng, genome[off:off + int(rng.integers(3 * n // 4, n + 1))], alphabet)
            if width in (40, 250):
                a, b, off = b, a, -off
            lo = off - width // 2 + int(rng.integers(-6, 6))
            pieces += [a, b]
            spec.append((at, len(a), at + len(a), len(b), lo, lo + width - 1))
            at += len(a) + len(b)
    kmer, spec = np.concatenate(pieces), np.asarray(spec, dtype=np.int64)
    want = [orc.banded_dp(kmer[b0:b0 + nx], kmer[b1:b1 + ny], int(lo), int(hi)) for b0, nx, b1, ny, lo, hi in spec]
    with switched_off():
        got = _run(lib, kmer, spec)
    for (x, sx), (y, sy) in zip(want, got):
        assert sx == sy and np.array_equal(x, y)
    assert all(int(nx) + int(ny) >= 4300 for _, nx, _, ny, _, _ in spec), [int(nx) + int(ny) for _, nx, _, ny, _, _ in spec]
    return len(spec)
