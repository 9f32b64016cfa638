"""The sharded job on two ranks (gloo) with the real stages of the library: same driver as the RCCL
path, only the transport differs.  Bit-exact against the single-process oracle, for LowHash0 and for
Align4 on the re-split candidate list.  Shared by the GPU test (both ranks on cuda:0 -- the GPU box has
one GPU) and its emulated pre-flight (library_path = the emulated build, tensors on the host)."""
import os
import tempfile

import numpy as np
import torch.multiprocessing as mp

from shasta_amd import abi, distributed
from tests import support


def _worker(rank, world, port, out_dir, seed, kw, library_path, transport="gloo"):
    import torch
    import torch.distributed as dist
    import shasta_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if transport == "nccl":
        torch.cuda.set_device(0)                 # (before the process group: RCCL binds the communicator to the current device)
    dist.init_process_group(transport, rank=rank, world_size=world)
    try:
        if library_path is None:
            torch.cuda.set_device(0)
            lib, device = shasta_amd.load(), "cuda:0"
        else:
            from shasta_amd import lib as libmod
            lib, device = libmod.Library(library_path), "cpu"      # emulated build: "device" memory is host memory
        toc, kmer, data7 = support.small_marker_set(n_reads=400, genome_markers=25000, seed=seed)
        flags = np.zeros(400, np.uint8)
        flags[[0, 7, 399]] = 1
        p = abi.default_lowhash0_params(**kw)
        with lib.context(0) as ctx:
            ctx.set_markers(toc, data7, flags)
            backend = distributed.HipBackend(ctx, device)
            boundaries = distributed.read_boundaries(toc, world)
            # (every other case with the candidates left on the device for the all-gather: lh_finish_on_device)
            out = distributed.lowhash0(backend, p, 400, boundaries, candidates_on_device=(seed % 2 == 1))
            everything = distributed.gather_candidates(out.candidates, device)
            lo, hi = distributed.candidate_slice(len(everything), rank, world)
            o = abi.default_align4_options(minAlignedMarkerCount=40)
            al = ctx.align4(everything[lo:hi], o, want_ordinals=True)
            np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
                     candidates=np.stack([everything["readId0"], everything["readId1"], everything["isSameStrand"]], axis=1),
                     statistics=out.statistics, high=out.high_frequency, total=out.total, histogram=out.histogram,
                     log2=np.asarray([out.log2_bucket_count]), slice=np.asarray([lo, hi]),
                     status=np.array(al.status), info=al.info_table(), ordinals=np.array(al.ordinals),
                     ordinals_toc=np.array(al.ordinals_toc), compressed=np.array(al.compressed_data),
                     compressed_toc=np.array(al.compressed_toc))
    finally:
        dist.destroy_process_group()


CASES = [
    (71, dict(minBucketSize=3, maxBucketSize=30, minFrequency=2)),
    (72, dict(minHashIterationCount=0, alignmentCandidatesPerRead=6.0, maxBucketSize=40)),      # dynamic control: iteration after iteration
    (73, dict(log2MinHashBucketCount=31, minHashIterationCount=5, hashFraction=0.03, minBucketSize=2, maxBucketSize=30, minFrequency=2)),
]


def two_ranks_equal_single_process_oracle(oracle_lib, seed, kw, library_path=None, port_base=29700, world=2, transport="gloo"):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port_base + seed, d, seed, kw, library_path, transport), nprocs=world, join=True)
        toc, kmer, data7 = support.small_marker_set(n_reads=400, genome_markers=25000, seed=seed)
        flags = np.zeros(400, np.uint8)
        flags[[0, 7, 399]] = 1
        ref = oracle_lib.lowhash0(toc, data7, flags, abi.default_lowhash0_params(**kw))
        assert len(ref.candidates) > 100
        o = abi.default_align4_options(minAlignedMarkerCount=40)
        ra = oracle_lib.align4_batch(toc, data7, ref.candidates, o, want_ordinals=True, threads=0)
        status, info, ordinals, compressed = [], [], [], []
        for rank in range(world):
            z = np.load(os.path.join(d, "rank%d.npz" % rank))
            assert np.array_equal(z["candidates"], ref.candidate_tuples())
            assert np.array_equal(z["statistics"], ref.statistics)
            assert np.array_equal(z["high"], ref.high_frequency)
            assert np.array_equal(z["total"], ref.total)
            assert np.array_equal(z["histogram"], ref.histogram)
            assert int(z["log2"][0]) == ref.log2_bucket_count
            status.append(z["status"]); info.append(z["info"]); ordinals.append(z["ordinals"]); compressed.append(z["compressed"])
        # Align4 results of the rank slices, concatenated in rank order = the single-process result.
        assert np.array_equal(np.concatenate(status) & 0x7f, ra.status & 0x7f)
        assert np.array_equal(np.concatenate(info), ra.info_table())
        assert np.array_equal(np.concatenate(ordinals), ra.ordinals)
        assert np.array_equal(np.concatenate(compressed), ra.compressed_data)
