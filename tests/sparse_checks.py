"""The sparse form of the banded alignment (shasta_amd/csrc/align4_sparse.hpp: the alignment from the matches inside the band,
answered only where the optimal chain of matches is unique) against the dense kernels and the oracle: the same results whether
the sparse path is on or off and under every compiled tie policy, most tasks of clean reads certified, tie-heavy and repeat-rich
tasks handed to the dense kernels.  Shared by the -m gpu tests and their pre-flight on the emulated build."""
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
        for name in ("sparse", "dense"):
            ctx.kernel_table_reset()
            if name == "dense":
                with switched_off():
                    got = ctx.align4(cand, o, want_ordinals=True)
            else:
                got = ctx.align4(cand, o, want_ordinals=True)
            table = ctx.kernel_table()
            rows[name] = sum(v["work"] for k, v in table.items() if k.startswith("bandedDpForwardKernel"))
            assert ("sparseChainKernel" in table) == (name == "sparse")
            ties = (want.status & 0x80) != 0
            if not ties.any():
                support.same_align(want, got)
            else:
                assert want.per_candidate(~ties) == got.per_candidate(~ties)
            assert got.dp_cell_count == want.dp_cell_count
    assert rows["dense"] == want.dp_cell_count
    return 1.0 - rows["sparse"] / max(1, rows["dense"])
