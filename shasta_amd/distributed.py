"""One LowHash0 + Align4 job sharded over several GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  The reference has no
counterpart -- it is a single shared-memory process -- so this module follows the reference's
*semantics* (src/LowHash0.cpp) and produces its exact output:

* reads are split into contiguous ranges balanced by marker count; every rank hashes its own
  range (no communication);
* bucket ids are owned in contiguous ranges: all-to-all(v) of the low-hash records -- of ALL
  iterations at once (16 bytes each) when their number is fixed, per iteration (12 bytes each)
  under the dynamic iteration control;
* pair keys are owned by the rank whose read range contains readId0: all-to-all(v) of the 8-byte
  keys (with their iteration tags in the one-pass form); the per-iteration counters, the bucket-size
  histograms and the per-read statistics are all-reduced once, after the last iteration;
* each rank's candidates are sorted and cover its readId0 range, so the concatenation in rank
  order is the reference's candidate list;
* Align4 candidates are independent: the candidate list is all-gathered and re-split into contiguous
  shares balanced by the markers they touch (sum of nx + ny).

The compute stages are behind a small backend interface (tensors in, tensors out): the product
backend is HipBackend (the C ABI's shasta_mi355x_lh_* entry points); the CPU tests plug in a numpy
backend to exercise this file's sharding and exchange logic under gloo.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import abi

SIZE_HISTOGRAM_BINS = 2048


def read_boundaries(toc, world):
    """world+1 read ids splitting the reads into contiguous ranges of about equal marker count."""
    toc = np.asarray(toc, dtype=np.uint64)
    read_count = (len(toc) - 1) // 2
    per_read_end = toc[2::2].astype(np.float64)          # markers up to and including each read
    total = float(toc[-1])
    b = [0]
    for r in range(1, world):
        b.append(int(np.searchsorted(per_read_end, total * r / world, side="left")))
    b.append(read_count)
    return np.maximum.accumulate(np.asarray(b, dtype=np.uint64))


class HipBackend:
    """The stages on this rank's GPU through the C ABI (include/shasta_mi355x.h, lh_* entry points)."""

    def __init__(self, ctx, device):
        self.ctx = ctx
        self.device = torch.device(device)

    def begin(self, params, rank, world, boundaries):
        return self.ctx.lh_begin(params, rank, world, boundaries)

    class _DeviceArray:
        """n elements at a device address, as torch reads them (__cuda_array_interface__): a VIEW of the library's buffer."""
        def __init__(self, ptr, n, typestr):
            self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}

    def _tensor_from(self, ptr, n, dtype):
        """The library's stage output as a tensor WITHOUT a copy: RCCL sends straight from the buffer the stage wrote (it stays
        valid until the context's next stage call, and the exchange is over by then).  Where torch cannot wrap a raw device
        address (a CPU-only build driving the emulated library), a copy as before."""
        n = int(n)
        if n == 0:
            return torch.empty(0, dtype=dtype, device=self.device)
        if self.device.type == "cuda" and not self._copies:
            try:
                return torch.as_tensor(self._DeviceArray(ptr, n, "<i4" if dtype == torch.int32 else "<i8"), device=self.device)
            except (TypeError, ValueError, RuntimeError) as e:      # (a torch that cannot wrap a raw address; remembered: every later call copies)
                self._copies = True
                import sys
                sys.stderr.write("shasta_amd.distributed: stage outputs are copied, not viewed (%s: %s)\n" % (type(e).__name__, str(e)[:200]))
        t = torch.empty(n, dtype=dtype, device=self.device)
        self.ctx.memcpy(t.data_ptr(), ptr, n * t.element_size(), 2)
        return t

    _copies = False

    def hash(self, iteration):
        offsets, keys, vals = self.ctx.lh_hash(iteration)
        n = int(offsets[-1])
        return offsets, self._tensor_from(keys, n, torch.int32), self._tensor_from(vals, n, torch.int64)

    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def buckets(self, keys, vals):
        self._sync()
        n = keys.numel()
        offsets, pk, used, hist, overflow = self.ctx.lh_buckets(keys.data_ptr() if n else 0, vals.data_ptr() if n else 0, n)
        m = int(offsets[-1])
        return offsets, self._tensor_from(pk, m, torch.int64), used, hist, overflow

    def merge(self, pair_keys, evaluate_now):
        self._sync()
        n = pair_keys.numel()
        return self.ctx.lh_merge(pair_keys.data_ptr() if n else 0, n, evaluate_now)

    def finish(self):
        return self.ctx.lh_finish()

    def finish_on_device(self):
        """finish() with the candidates left on the device: an int32 tensor of triplets (readId0, readId1, isSameStrand in the
        low byte), copied there from the library's buffer."""
        ptr, count, stats, high, total = self.ctx.lh_finish_on_device()
        # (A copy of its own: the caller keeps this tensor, the library's buffer is reused by the context's next job.  The views
        # of the stage outputs above live only until the exchange that follows them.)
        return self._tensor_from(ptr, 3 * count, torch.int32).clone(), stats, high, total

    # All iterations in one pass (fixed minHashIterationCount): one call of each per job.
    def one_pass_fits(self):
        return self.ctx.lh_one_pass_fits()

    def hash_all(self):
        offsets, keys, vals = self.ctx.lh_hash_all()
        n = int(offsets[-1])
        return offsets, self._tensor_from(keys, n, torch.int64), self._tensor_from(vals, n, torch.int64)

    def buckets_all(self, keys, vals):
        self._sync()
        n = keys.numel()
        offsets, pk, tags, used, hist, overflow = self.ctx.lh_buckets_all(keys.data_ptr() if n else 0, vals.data_ptr() if n else 0, n)
        m = int(offsets[-1])
        return offsets, self._tensor_from(pk, m, torch.int64), self._tensor_from(tags, m, torch.int32), used, hist, overflow

    def merge_all(self, pair_keys, tags):
        self._sync()
        n = pair_keys.numel()
        self.ctx.lh_merge_all(pair_keys.data_ptr() if n else 0, tags.data_ptr() if n else 0, n)


def _comm_device(tensor_device):
    """Collectives run on the tensors' device with nccl (RCCL) and on the host with gloo."""
    return tensor_device if dist.get_backend() == "nccl" else torch.device("cpu")


def exchange(tensors, send_offsets, group=None):
    """all-to-all(v): tensors share the split send_offsets (world+1); returns the received tensors."""
    world = dist.get_world_size(group)
    send_counts = np.diff(np.asarray(send_offsets, dtype=np.int64))
    assert len(send_counts) == world
    home = tensors[0].device
    comm = _comm_device(home)
    sc = torch.tensor(send_counts, dtype=torch.int64, device=comm)
    rc = torch.empty(world, dtype=torch.int64, device=comm)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = [int(x) for x in rc.tolist()]
    # The data collectives of all the tensors are in flight together (one wait at the end).
    out, pending = [], []
    for t in tensors:
        src = t.to(comm).contiguous()
        dst = torch.empty(sum(recv_counts), dtype=t.dtype, device=comm)
        pending.append(dist.all_to_all_single(dst, src, output_split_sizes=recv_counts, input_split_sizes=[int(x) for x in send_counts], group=group, async_op=True))
        out.append(dst)
    for work in pending:
        work.wait()
    return [dst.to(home) for dst in out]


def all_gather_padded(tensor, counts, group=None):
    """All-gather of 1-D tensors of different lengths (counts[r] elements from rank r), concatenated."""
    world = dist.get_world_size(group)
    home = tensor.device
    comm = _comm_device(home)
    pad = max(int(c) for c in counts)
    mine = torch.zeros(pad, dtype=tensor.dtype, device=comm)
    mine[:tensor.numel()] = tensor.to(comm)
    parts = [torch.empty(pad, dtype=tensor.dtype, device=comm) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return torch.cat([parts[r][:int(counts[r])].to(home) for r in range(world)])


def all_reduce_sum_u64(array, device, group=None):
    """Sum of uint64 numpy arrays over the ranks (two's complement wrap-around = uint64 arithmetic)."""
    a = np.ascontiguousarray(array, dtype=np.uint64)
    t = torch.from_numpy(a.view(np.int64).copy()).to(_comm_device(torch.device(device)))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy().view(np.uint64).reshape(a.shape)


def _every_rank_agrees(flag, device, group=None):
    """True if `flag` is true on every rank (one small all-reduce)."""
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=_comm_device(torch.device(device)))
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


class LowHash0Result:
    def __init__(self):
        self.candidates = None          # this rank's (sorted) share
        self.statistics = None          # global (all-reduced)
        self.high_frequency = None
        self.total = None
        self.histogram = None
        self.log2_bucket_count = 0


def lowhash0(backend, params, read_count, boundaries, group=None, candidates_on_device=False):
    """Runs the job; every rank gets the global counters/statistics and its own share of the candidates (a numpy array of
    12-byte pairs; with candidates_on_device and a backend that can, an int32 tensor of triplets on the backend's device --
    candidate_share / gather_candidates take either).
    Per MinHash iteration the ranks only exchange DATA (two all-to-all steps, each preceded by its counts); every
    reduction -- per-iteration counters, bucket histograms, per-read statistics -- happens once, after the last
    iteration.  Only minHashIterationCount = 0 needs the global high-frequency count after every iteration
    (src/LowHash0.cpp:137-149): then the keys are evaluated and that one number all-reduced per iteration."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    device = backend.device
    log2 = backend.begin(params, rank, world, boundaries)
    bucket_count = 1 << log2
    dynamic = params.minHashIterationCount == 0
    used_rows, hist_rows, overflow_lists = [], [], []
    high_frequency = 0
    iteration = 0
    # A fixed number of iterations: all of them in ONE pass -- the markers hashed once under every seed, one exchange of the
    # records of all iterations, one of their pair keys: 2 data exchanges per job instead of 2 per iteration (per-link xGMI
    # bandwidth wants few large collectives).  SHASTA_MI355X_LOWHASH_ONE_PASS=0: iteration after iteration, as with the dynamic control.
    one_pass = (not dynamic and 1 <= int(params.minHashIterationCount) <= 4096 and world <= 256 and hasattr(backend, "hash_all")
                and os.environ.get("SHASTA_MI355X_LOWHASH_ONE_PASS", "1") != "0")
    if one_pass:
        # ... and only if the records of all iterations fit one sort on EVERY rank (each asks its own context; a rank that went
        # ahead alone would leave the others waiting in the exchange): otherwise iteration after iteration.
        one_pass = _every_rank_agrees(backend.one_pass_fits() if hasattr(backend, "one_pass_fits") else True, device, group)
    if one_pass:
        offsets, keys, vals = backend.hash_all()
        keys, vals = exchange([keys, vals], offsets, group)                       # C1: records of all iterations to bucket owners
        if not _every_rank_agrees(int(keys.shape[0] if hasattr(keys, "shape") else len(keys)) < (1 << 32) - 1, device, group):
            raise RuntimeError("LowHash0: a rank received more low-hash records of all iterations than one sort takes (2^32): "
                               "run with SHASTA_MI355X_LOWHASH_ONE_PASS=0")
        offsets, pair_keys, tags, used, hist, overflow = backend.buckets_all(keys, vals)
        pair_keys, tags = exchange([pair_keys, tags], offsets, group)             # C2: pair keys + iteration tags to readId0 owners
        backend.merge_all(pair_keys, tags)
        iteration = int(params.minHashIterationCount)
        overflow = np.asarray(overflow, dtype=np.uint64)
        for t in range(iteration):
            used_rows.append(int(used[t]))
            hist_rows.append(np.asarray(hist[t], dtype=np.uint64))
            overflow_lists.append([int(e & np.uint64(0xffffffff)) for e in overflow[(overflow >> np.uint64(32)) == np.uint64(t)]])
    while not one_pass:
        # Iteration control, src/LowHash0.cpp:136-157 (on the global counter: every rank decides alike).
        if dynamic:
            if 2.0 * float(high_frequency) / float(read_count) >= params.alignmentCandidatesPerRead:
                break
        elif iteration == params.minHashIterationCount:
            break
        offsets, keys, vals = backend.hash(iteration)
        keys, vals = exchange([keys, vals], offsets, group)                       # C1: records to bucket owners
        offsets, pair_keys, used, hist, overflow = backend.buckets(keys, vals)
        (pair_keys,) = exchange([pair_keys], offsets, group)                      # C2: pair keys to readId0 owners
        high, _ = backend.merge(pair_keys, dynamic)
        if dynamic:
            high_frequency = int(all_reduce_sum_u64(np.asarray([high], dtype=np.uint64), device, group)[0])
        used_rows.append(used)
        hist_rows.append(np.asarray(hist, dtype=np.uint64))
        overflow_lists.append(np.asarray(overflow, dtype=np.uint32).tolist())
        iteration += 1
    if candidates_on_device and hasattr(backend, "finish_on_device"):
        candidates, stats, high_rows, total_rows = backend.finish_on_device()
    else:
        candidates, stats, high_rows, total_rows = backend.finish()
    iterations = iteration
    # One reduction for everything: [high | total | bucketsUsed | overflow counts | histograms] per iteration, then the statistics.
    packed = np.concatenate([np.asarray(high_rows, dtype=np.uint64), np.asarray(total_rows, dtype=np.uint64),
                             np.asarray(used_rows, dtype=np.uint64), np.asarray([len(o) for o in overflow_lists], dtype=np.uint64)]
                            + hist_rows) if iterations else np.zeros(0, dtype=np.uint64)
    packed = all_reduce_sum_u64(packed, device, group)
    high_all, total_all = packed[:iterations], packed[iterations:2 * iterations]
    used_all, overflow_all = packed[2 * iterations:3 * iterations], packed[3 * iterations:4 * iterations]
    hist_all = packed[4 * iterations:].reshape(iterations, SIZE_HISTOGRAM_BINS) if iterations else np.zeros((0, SIZE_HISTOGRAM_BINS), np.uint64)
    # Bucket sizes beyond the histogram bins are rare: gather the lists only when there are any.
    lists = [[[] for _ in range(iterations)]] * world
    if int(overflow_all.sum()):
        lists = [None] * world
        dist.all_gather_object(lists, overflow_lists, group=group)
    histogram_rows = []
    for it in range(iterations):
        rows = {}
        if bucket_count > int(used_all[it]):
            rows[0] = bucket_count - int(used_all[it])
        for s in np.nonzero(hist_all[it][1:])[0] + 1:
            rows[int(s)] = int(hist_all[it][s])
        for per_rank in lists:
            for s in per_rank[it]:
                rows[int(s)] = rows.get(int(s), 0) + 1
        for s in sorted(rows):
            histogram_rows.append((it, s, rows[s]))
    out = LowHash0Result()
    out.candidates = candidates
    out.statistics = all_reduce_sum_u64(stats, device, group)
    out.high_frequency = np.asarray(high_all, dtype=np.uint64)
    out.total = np.asarray(total_all, dtype=np.uint64)
    out.histogram = np.asarray(histogram_rows, dtype=np.uint64).reshape(-1, 3)
    out.log2_bucket_count = log2
    return out


def gather_candidates(local_candidates, device="cpu", group=None):
    """The global candidate list (rank order = the reference's order) on every rank.  Candidates are
    12-byte records; they travel as int32 triplets in one padded all-gather (24 MB per million)."""
    world = dist.get_world_size(group)
    home = torch.device(device)
    flat, counts = _flat_candidates_and_counts(local_candidates, home, group)
    gathered = all_gather_padded(flat, [3 * c for c in counts], group)
    return gathered.cpu().numpy().view(abi.PAIR_DTYPE).reshape(-1)


def _flat_candidates_and_counts(local_candidates, home, group):
    """A rank's candidates as an int32 tensor of triplets on `home` (from the numpy array of pairs, or the tensor
    lowhash0(candidates_on_device=True) returned, as it is) and every rank's number of candidates."""
    world = dist.get_world_size(group)
    if isinstance(local_candidates, torch.Tensor):
        flat = local_candidates.to(home)
    else:
        local = np.ascontiguousarray(local_candidates, dtype=abi.PAIR_DTYPE).view(np.int32).reshape(-1)
        flat = torch.from_numpy(local if local.flags.writeable else local.copy()).to(home)
    comm = _comm_device(home)
    mine = torch.tensor([flat.numel() // 3], dtype=torch.int64, device=comm)
    parts = torch.zeros(world, dtype=torch.int64, device=comm)
    dist.all_gather_into_tensor(parts, mine, group=group)
    return flat, [int(c) for c in parts.tolist()]


def candidate_share(local_candidates, device="cpu", group=None, toc=None):
    """This rank's share of the global candidate list for the aligner, and the list's length.  The list is
    all-gathered on the device; only the share crosses to the host.  With `toc` (Markers.toc of all reads) the
    contiguous shares are balanced by the markers they touch, sum of nx + ny: reads differ in length by
    an order of magnitude and the aligner's work follows them; without it the split is even."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    home = torch.device(device)
    flat, counts = _flat_candidates_and_counts(local_candidates, home, group)
    total = sum(counts)
    gathered = all_gather_padded(flat, [3 * c for c in counts], group)
    if toc is None:
        lo, hi = candidate_slice(total, rank, world)
        return gathered[3 * lo:3 * hi].cpu().numpy().view(abi.PAIR_DTYPE).reshape(-1), total
    # The weights and their prefix sums where the gathered list is (the device): with world ranks the list is world times a
    # rank's own, and the numpy form of this -- two gathers and a cumulative sum over ALL candidates on every rank's host, after
    # a device-to-host copy of all of them -- was 17 ms per step with one rank of 2 M candidates.
    lo, hi = _slice_by_markers_on(gathered, _toc_on(toc, home), rank, world)
    return gathered[3 * lo:3 * hi].cpu().numpy().view(abi.PAIR_DTYPE).reshape(-1), total


_toc_cache = {}


def _toc_on(toc, device):
    """Markers.toc as an int64 tensor on `device` (kept: the same array is handed in at every step)."""
    # (keyed on the array AND on what it says -- length, total, a checksum of a sample -- so that a toc changed in place is seen)
    sample = np.asarray(toc[:: max(1, len(toc) // 4096)], dtype=np.uint64)
    key = (id(toc), str(device), len(toc), int(toc[-1]) if len(toc) else 0, int(sample.sum(dtype=np.uint64)))
    hit = _toc_cache.get(key)
    if hit is None or hit[0] is not toc:
        _toc_cache.clear()
        hit = (toc, torch.from_numpy(np.ascontiguousarray(toc, dtype=np.uint64).view(np.int64)).to(device))
        _toc_cache[key] = hit
    return hit[1]


def _slice_by_markers_on(flat, toc, rank, world):
    """candidate_slice_by_markers on the tensor of int32 triplets (readId0, readId1, isSameStrand in the low byte): the same
    cuts, computed where the tensor lives."""
    n = flat.numel() // 3
    if n == 0:
        return 0, 0
    c = flat.view(-1, 3).to(torch.int64)
    sizes = toc[1:] - toc[:-1]
    o0 = 2 * c[:, 0]
    o1 = 2 * c[:, 1] + ((c[:, 2] & 0xff) == 0).to(torch.int64)
    prefix = torch.cumsum(sizes[o0] + sizes[o1] + 64, dim=0)
    last = int(prefix[-1].item())
    cuts = [0]
    for r in range(1, world):
        target = int(float(last) * float(r) / float(world))
        # first index i of [0, prefix...] with value >= target  (numpy.searchsorted(..., side="left") on the array with a leading 0)
        at = 0 if target <= 0 else int(torch.searchsorted(prefix, torch.tensor([target], dtype=torch.int64, device=prefix.device), right=False).item()) + 1
        cuts.append(min(max(at, cuts[-1]), n))
    cuts.append(n)
    return cuts[rank], cuts[rank + 1]


def candidate_slice(count, rank, world):
    """Even split of the candidate list (candidates are independent)."""
    return (count * rank) // world, (count * (rank + 1)) // world


def candidate_slice_by_markers(candidates, toc, rank, world):
    """Contiguous split with equal shares of sum(nx + ny + 64) -- the same rule as the in-process group
    (shasta_amd/csrc/multi.hip, Group::alignRun)."""
    toc = np.asarray(toc, dtype=np.int64)
    sizes = np.diff(toc)
    o0 = 2 * candidates["readId0"].astype(np.int64)
    o1 = 2 * candidates["readId1"].astype(np.int64) + (candidates["isSameStrand"] == 0).astype(np.int64)
    prefix = np.concatenate([[0], np.cumsum(sizes[o0] + sizes[o1] + 64)])
    cuts = [0]
    for r in range(1, world):
        target = int(float(prefix[-1]) * float(r) / float(world))
        cuts.append(min(max(int(np.searchsorted(prefix, target, side="left")), cuts[-1]), len(candidates)))
    cuts.append(len(candidates))
    return cuts[rank], cuts[rank + 1]
