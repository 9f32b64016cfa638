"""A numpy implementation of the LowHash0 stage interface of shasta_amd/distributed.py (test
infrastructure: lets the CPU tests drive the sharding / exchange logic under gloo without a GPU).
Semantics follow src/LowHash0.cpp exactly as the HIP stages do; MurmurHash64A comes from the oracle."""
import numpy as np
import torch


class NumpyBackend:
    def __init__(self, toc, kmer_ids, flags, oracle):
        self.toc = np.asarray(toc, dtype=np.int64)
        self.kmer = np.asarray(kmer_ids, dtype=np.uint32)
        self.read_count = (len(self.toc) - 1) // 2
        self.flags = np.zeros(self.read_count, np.uint8) if flags is None else np.asarray(flags, np.uint8)
        self.oracle = oracle
        self.device = torch.device("cpu")

    def begin(self, params, rank, world, boundaries):
        self.p, self.rank, self.world = params, rank, world
        self.boundaries = np.asarray(boundaries, dtype=np.int64)
        M = int(self.toc[-1])
        estimate = int(params.hashFraction * float(M))
        log2_estimate = estimate.bit_length()
        log2 = int(params.log2MinHashBucketCount) or 5 + log2_estimate
        log2 = min(log2, 31)
        self.log2 = log2
        self.mask = (1 << log2) - 1
        self.threshold = int(np.uint64(np.float64(params.hashFraction) * np.float64(np.iinfo(np.uint64).max)))
        self.read_bits = max(1, int(self.read_count - 1).bit_length())
        self.stats = np.zeros((self.read_count, 3), np.uint64)
        self.table = {}
        self.high_rows, self.total_rows = [], []
        self.min_frequency = min(int(params.minFrequency), 0x10000)
        return log2

    def _owner_offsets(self, sorted_keys, bounds):
        return np.searchsorted(sorted_keys, bounds, side="left").astype(np.uint64)

    def hash(self, iteration):
        m = int(self.p.m)
        keys, vals = [], []
        for r in range(int(self.boundaries[self.rank]), int(self.boundaries[self.rank + 1])):
            if self.flags[r] & 1:
                continue
            for strand in (0, 1):
                o = 2 * r + strand
                k = self.kmer[self.toc[o]:self.toc[o + 1]]
                if len(k) < m:
                    continue
                h = self.oracle.hash_windows(k, m, iteration)
                h = h[h < np.uint64(self.threshold)]
                keys.append((h & np.uint64(self.mask)).astype(np.uint32))
                vals.append((h & np.uint64(0xffffffff00000000)) | np.uint64(o))
        keys = np.concatenate(keys) if keys else np.zeros(0, np.uint32)
        vals = np.concatenate(vals) if vals else np.zeros(0, np.uint64)
        order = np.argsort(keys, kind="stable")
        keys, vals = keys[order], vals[order]
        bucket_count = 1 << self.log2
        bounds = np.asarray([(r * bucket_count + self.world - 1) // self.world for r in range(self.world + 1)], dtype=np.uint64)
        offsets = self._owner_offsets(keys.astype(np.uint64), bounds)
        offsets[0], offsets[-1] = 0, len(keys)
        return offsets, torch.from_numpy(keys.view(np.int32).copy()), torch.from_numpy(vals.view(np.int64).copy())

    def buckets(self, keys, vals):
        keys = keys.numpy().view(np.uint32)
        vals = vals.numpy().view(np.uint64)
        order = np.argsort(keys, kind="stable")
        keys, vals = keys[order], vals[order]
        hist = np.zeros(2048, np.uint64)
        overflow = []
        pair_keys = []
        min_b, max_b = int(self.p.minBucketSize), int(self.p.maxBucketSize)
        starts = np.flatnonzero(np.r_[True, keys[1:] != keys[:-1]]) if len(keys) else np.zeros(0, np.int64)
        ends = np.r_[starts[1:], len(keys)] if len(keys) else starts
        for b, e in zip(starts, ends):
            size = int(e - b)
            if size < 2048:
                hist[size] += 1
            else:
                overflow.append(size)
            v = vals[b:e]
            oriented = (v & np.uint64(0xffffffff)).astype(np.int64)
            reads = oriented >> 1
            cls = 0 if size < min_b else (2 if size > max_b else 1)
            np.add.at(self.stats[:, cls], reads, 1)
            if max(2, min_b) <= size <= max_b:
                high = v >> np.uint64(32)
                for a in range(size):
                    for c in range(size):
                        if high[a] == high[c] and reads[a] < reads[c]:
                            strand_bit = (int(oriented[a]) ^ int(oriented[c])) & 1
                            pair_keys.append((int(reads[a]) << (self.read_bits + 1)) | (int(reads[c]) << 1) | strand_bit)
        pk = np.sort(np.asarray(pair_keys, dtype=np.uint64))
        bounds = (self.boundaries.astype(np.uint64) << np.uint64(self.read_bits + 1))
        offsets = self._owner_offsets(pk, bounds)
        offsets[0], offsets[-1] = 0, len(pk)
        return offsets, torch.from_numpy(pk.view(np.int64).copy()), len(starts), hist, np.asarray(overflow, np.uint32)

    # The job with all iterations in one pass: the same stages run iteration by iteration here, packed as the HIP stages
    # pack them (keys owner << 56 | iteration << 32 | bucket id, each owner's records contiguous; pair keys with tags).
    def one_pass_fits(self):
        # (SHASTA_TEST_NO_ONE_PASS_ON_RANK=<r>: rank r alone says no -- every rank must then run iteration after iteration.)
        import os
        self.asked_one_pass = True
        return os.environ.get("SHASTA_TEST_NO_ONE_PASS_ON_RANK") != str(self.rank)

    def hash_all(self):
        self.ran_one_pass = True
        iterations = int(self.p.minHashIterationCount)
        per_owner_keys = [[] for _ in range(self.world)]
        per_owner_vals = [[] for _ in range(self.world)]
        for t in range(iterations):
            offsets, keys, vals = self.hash(t)
            keys = keys.numpy().view(np.uint32).astype(np.uint64)
            vals = vals.numpy().view(np.uint64)
            for r in range(self.world):
                lo, hi = int(offsets[r]), int(offsets[r + 1])
                per_owner_keys[r].append(keys[lo:hi] | (np.uint64(t) << np.uint64(32)) | (np.uint64(r) << np.uint64(56)))
                per_owner_vals[r].append(vals[lo:hi])
        sizes = [sum(len(k) for k in per_owner_keys[r]) for r in range(self.world)]
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        keys = np.concatenate([k for r in range(self.world) for k in per_owner_keys[r]]) if sum(sizes) else np.zeros(0, np.uint64)
        vals = np.concatenate([v for r in range(self.world) for v in per_owner_vals[r]]) if sum(sizes) else np.zeros(0, np.uint64)
        return offsets, torch.from_numpy(keys.view(np.int64).copy()), torch.from_numpy(vals.view(np.int64).copy())

    def buckets_all(self, keys, vals):
        iterations = int(self.p.minHashIterationCount)
        keys = keys.numpy().view(np.uint64)
        vals = vals.numpy().view(np.uint64)
        of = (keys >> np.uint64(32)) & np.uint64(0xffffff)
        used, hists, overflow, pair_keys, tags = [], [], [], [], []
        for t in range(iterations):
            sel = of == np.uint64(t)
            k32 = (keys[sel] & np.uint64(0xffffffff)).astype(np.uint32)
            _, pk, u, hist, over = self.buckets(torch.from_numpy(k32.view(np.int32).copy()), torch.from_numpy(vals[sel].view(np.int64).copy()))
            pk = pk.numpy().view(np.uint64)
            used.append(u); hists.append(hist)
            overflow += [(t << 32) | int(s) for s in over]
            pair_keys.append(pk); tags.append(np.full(len(pk), t, np.uint32))
        pk = np.concatenate(pair_keys) if pair_keys else np.zeros(0, np.uint64)
        tg = np.concatenate(tags) if tags else np.zeros(0, np.uint32)
        order = np.argsort(pk, kind="stable")
        pk, tg = pk[order], tg[order]
        bounds = (self.boundaries.astype(np.uint64) << np.uint64(self.read_bits + 1))
        offsets = self._owner_offsets(pk, bounds)
        offsets[0], offsets[-1] = 0, len(pk)
        return (offsets, torch.from_numpy(pk.view(np.int64).copy()), torch.from_numpy(tg.view(np.int32).copy()),
                np.asarray(used, np.uint64), np.asarray(hists, np.uint64).reshape(iterations, 2048), np.asarray(overflow, np.uint64))

    def merge_all(self, pair_keys, tags):
        pk = pair_keys.numpy().view(np.uint64)
        tg = tags.numpy().view(np.uint32)
        for t in range(int(self.p.minHashIterationCount)):
            self.merge(torch.from_numpy(pk[tg == t].view(np.int64).copy()), False)

    def merge(self, pair_keys, evaluate_now):
        # The reference folds the iteration's pairs into a uint16 frequency (src/LowHash0.hpp:116: additions wrap).
        for k in pair_keys.numpy().view(np.uint64).tolist():
            self.table[k] = (self.table.get(k, 0) + 1) & 0xffff
        self.high_rows.append(sum(1 for c in self.table.values() if c >= self.min_frequency))
        self.total_rows.append(len(self.table))
        return (self.high_rows[-1], self.total_rows[-1]) if evaluate_now else (0, 0)

    def finish(self):
        from shasta_amd import abi
        keys = sorted(k for k, c in self.table.items() if c >= self.min_frequency)
        r0 = [k >> (self.read_bits + 1) for k in keys]
        r1 = [(k >> 1) & ((1 << self.read_bits) - 1) for k in keys]
        same = [0 if (k & 1) else 1 for k in keys]
        return (abi.make_pairs(np.asarray(r0, np.uint32), np.asarray(r1, np.uint32), np.asarray(same, np.uint8)), self.stats,
                np.asarray(self.high_rows, np.uint64), np.asarray(self.total_rows, np.uint64))
