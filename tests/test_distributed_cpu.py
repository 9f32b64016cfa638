"""world_size-2 gloo tests of the multi-GPU driver (shasta_amd/distributed.py) on CPU: the stages
come from a numpy backend (tests/dist_support.py), so what is exercised is the sharding, the two
all-to-all exchanges, the all-reduces and the candidate gathering.  The result must equal the
single-process oracle bit for bit."""
import os
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

from shasta_amd import abi, distributed
from tests import support


def _worker(rank, world, port, out_dir, seed, kw):
    import torch.distributed as dist
    from oracle import bindings
    from tests import dist_support
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        toc, kmer, data7 = support.small_marker_set(n_reads=120, genome_markers=8000, seed=seed)
        flags = np.zeros(120, np.uint8)
        flags[[3, 77]] = 1
        p = abi.default_lowhash0_params(**kw)
        backend = dist_support.NumpyBackend(toc, kmer, flags, bindings.OracleLib())
        boundaries = distributed.read_boundaries(toc, world)
        out = distributed.lowhash0(backend, p, 120, boundaries)
        # One rank that cannot take all iterations in one pass (SHASTA_TEST_NO_ONE_PASS_ON_RANK) sends every rank down the
        # iteration-by-iteration path: the decision is collective.
        if p.minHashIterationCount != 0:
            assert getattr(backend, "asked_one_pass", False)
            assert getattr(backend, "ran_one_pass", False) == (os.environ.get("SHASTA_TEST_NO_ONE_PASS_ON_RANK") is None)
        everything = distributed.gather_candidates(out.candidates)
        lo, hi = distributed.candidate_slice(len(everything), rank, world)
        share, total = distributed.candidate_share(out.candidates)
        assert total == len(everything) and np.array_equal(share, everything[lo:hi])
        # The split by markers: contiguous, covers everything, and its shares of sum(nx + ny) differ by less than one candidate's worth.
        blo, bhi = distributed.candidate_slice_by_markers(everything, toc, rank, world)
        balanced, total2 = distributed.candidate_share(out.candidates, toc=toc)
        assert total2 == total and np.array_equal(balanced, everything[blo:bhi])
        cuts = [distributed.candidate_slice_by_markers(everything, toc, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == len(everything) and all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
                 candidates=np.stack([everything["readId0"], everything["readId1"], everything["isSameStrand"]], axis=1),
                 local=np.stack([out.candidates["readId0"], out.candidates["readId1"], out.candidates["isSameStrand"]], axis=1),
                 statistics=out.statistics, high=out.high_frequency, total=out.total, histogram=out.histogram,
                 log2=np.asarray([out.log2_bucket_count]), boundaries=boundaries, slice=np.asarray([lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,seed,kw", [
    (2, 61, dict(minBucketSize=2, maxBucketSize=30, minFrequency=2)),
    (2, 62, dict(m=3, hashFraction=0.05, minHashIterationCount=0, alignmentCandidatesPerRead=6.0, maxBucketSize=40)),
    (3, 63, dict(hashFraction=0.03, log2MinHashBucketCount=14, minFrequency=1, minHashIterationCount=4)),
    (4, 64, dict(minBucketSize=2, maxBucketSize=30, minFrequency=2)),
    (8, 65, dict(m=3, hashFraction=0.04, minBucketSize=2, maxBucketSize=40, minFrequency=1, minHashIterationCount=5)),     # the node's eight GPUs: 15 reads per rank
])
def test_sharded_lowhash0_equals_single_process(oracle_lib, world, seed, kw):
    port = 29600 + seed
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, d, seed, kw), nprocs=world, join=True)
        toc, kmer, data7 = support.small_marker_set(n_reads=120, genome_markers=8000, seed=seed)
        flags = np.zeros(120, np.uint8)
        flags[[3, 77]] = 1
        ref = oracle_lib.lowhash0(toc, data7, flags, abi.default_lowhash0_params(**kw))
        assert len(ref.candidates) > 0
        covered = 0
        for rank in range(world):
            z = np.load(os.path.join(d, "rank%d.npz" % rank))
            assert np.array_equal(z["candidates"], ref.candidate_tuples())        # gathered list = the reference's order
            assert np.array_equal(z["statistics"], ref.statistics)
            assert np.array_equal(z["high"], ref.high_frequency)
            assert np.array_equal(z["total"], ref.total)
            assert np.array_equal(z["histogram"], ref.histogram)
            assert int(z["log2"][0]) == ref.log2_bucket_count
            # A rank's own candidates lie in its read range; the Align4 slices tile the list.
            b = z["boundaries"]
            if len(z["local"]):
                assert z["local"][:, 0].min() >= b[rank] and z["local"][:, 0].max() < b[rank + 1]
            lo, hi = z["slice"]
            assert lo == covered
            covered = hi
        assert covered == len(ref.candidates)


def test_one_rank_that_cannot_take_one_pass_sends_every_rank_down_the_other_path(oracle_lib, monkeypatch):
    monkeypatch.setenv("SHASTA_TEST_NO_ONE_PASS_ON_RANK", "1")
    test_sharded_lowhash0_equals_single_process(oracle_lib, 3, 66, dict(hashFraction=0.03, minFrequency=1, minHashIterationCount=4, minBucketSize=2, maxBucketSize=30))


def test_read_boundaries_balance_markers():
    toc, kmer, _ = support.small_marker_set(n_reads=200, genome_markers=12000, seed=7)
    for world in (1, 2, 4, 8):
        b = distributed.read_boundaries(toc, world)
        assert b[0] == 0 and b[-1] == 200 and np.all(np.diff(b.astype(np.int64)) >= 0)
        per_rank = np.diff(np.asarray(toc, np.int64)[2 * b.astype(np.int64)])
        assert per_rank.max() <= 1.5 * per_rank.mean() + 4000
