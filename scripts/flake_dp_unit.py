"""The dense DP kernels alone (the unit seam shasta_mi355x_banded_dp_many, sparse path off), on the tasks of the method-3 calls in which round 6
reproduced round 5's difference, the SAME task list again and again: the tasks' order inside a length bin and with it the tasks that share a
wavefront (a bundle) vary from run to run with the atomics that build the list, so a result that depends on a task's partners shows as a
difference between two runs -- whatever contexts, streams and the method-3 driver around the kernels do.
    python scripts/flake_dp_unit.py <repeats>      (FLAKE_LIBRARY=<path>: another build)"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["SHASTA_MI355X_SPARSE_DP"] = "0"
import numpy as np
import torch   # noqa: F401
import shasta_amd
from shasta_amd import abi, lib as libmod
from oracle import bindings
from tests import config_value_checks as cv


def tasks_of(orc, k, seed, rng):
    toc, kmer, data7 = cv.marker_set(k, 160, 9000, seed=seed, mean_markers=900.0, min_markers=300)
    p = abi.default_lowhash0_params(hashFraction=0.05, **cv.MAY2022_LOWHASH)
    cand = orc.lowhash0(toc, data7, None, p).candidates[:400]
    o3 = abi.default_align3_options(k=k, minAlignedFraction=0.4)
    x = orc.align3_batch(toc, data7, cand, o3, want_ordinals=True, threads=0)
    toc = toc.astype(np.int64)
    spec, pieces, at = [], [], 0
    for i in range(len(cand)):
        o0, o1 = 2 * int(cand["readId0"][i]), 2 * int(cand["readId1"][i]) + (0 if cand["isSameStrand"][i] else 1)
        a, b = kmer[toc[o0]:toc[o0 + 1]], kmer[toc[o1]:toc[o1 + 1]]
        pairs = x.ordinals_of(i)
        if len(pairs) == 0:
            continue
        off = pairs[:, 0].astype(np.int64) - pairs[:, 1].astype(np.int64)
        # Step 2 as the reference bands it: the offsets of the (down-sampled) alignment +- bandExtend, clipped to the matrix.
        lo, hi = max(int(off.min()) - 10, -len(b)), min(int(off.max()) + 10, len(a))
        pieces += [a, b]
        spec.append((at, len(a), at + len(a), len(b), lo, hi))
        at += len(a) + len(b)
        # Step 1 as the reference runs it: one marker in ten, every diagonal of the small matrix.
        ka, kb = a[rng.random(len(a)) < 0.1], b[rng.random(len(b)) < 0.1]
        if len(ka) and len(kb):
            pieces += [ka, kb]
            spec.append((at, len(ka), at + len(ka), len(kb), -len(kb), len(ka)))
            at += len(ka) + len(kb)
    return np.concatenate(pieces), np.asarray(spec, dtype=np.int64)


def main():
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    path = os.environ.get("FLAKE_LIBRARY")
    lib = libmod.Library(path) if path else shasta_amd.load()
    orc = bindings.OracleLib()
    rng = np.random.default_rng(3)
    sets = [tasks_of(orc, 14, 154, rng), tasks_of(orc, 16, 156, rng)]
    first, bad, t0 = [None, None], 0, time.time()
    threads = int(os.environ.get("FLAKE_THREADS", "1"))
    if threads > 1:
        # Several host threads, each making the same calls on the same device at the same time (ctypes releases the interpreter lock for the
        # call): what two contexts of a device list, or the aligner's own worker threads, do to the runtime and the device.
        import threading
        for s, (kmer, spec) in enumerate(sets):
            first[s] = lib.banded_dp_many(kmer, spec[:, 0], spec[:, 1], spec[:, 2], spec[:, 3], spec[:, 4], spec[:, 5])
        counts = [0] * threads

        def work(me):
            for it in range(repeats):
                for s, (kmer, spec) in enumerate(sets):
                    got = lib.banded_dp_many(kmer, spec[:, 0], spec[:, 1], spec[:, 2], spec[:, 3], spec[:, 4], spec[:, 5])
                    for t, ((x, sx), (y, sy)) in enumerate(zip(first[s], got)):
                        if sx != sy or not np.array_equal(x, y):
                            counts[me] += 1
                            d = int(np.sum(x != y)) if x.shape == y.shape else -1
                            print("thread", me, "repeat", it, "set", s, "task", t, "spec", spec[t].tolist(), "differs from the first run: scores", sx, sy, "pairs", len(x), len(y), "values that differ", d, flush=True)
        pool = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
        for th in pool:
            th.start()
        for th in pool:
            th.join()
        print("library %s: %d threads x repeats %d x (%d + %d tasks), differences from the first run %d, %.0f s" % (path or "(the product)", threads, repeats, len(sets[0][1]), len(sets[1][1]), sum(counts), time.time() - t0))
        return
    for it in range(repeats):
        for s, (kmer, spec) in enumerate(sets):
            if os.environ.get("FLAKE_PERMUTE") == "1" and first[s] is not None:
                # (the emulated build's atomics always run in the same order: the task list in another order instead, so that bundles differ)
                order = rng.permutation(len(spec))
                q = spec[order]
                shuffled = lib.banded_dp_many(kmer, q[:, 0], q[:, 1], q[:, 2], q[:, 3], q[:, 4], q[:, 5])
                got = [None] * len(spec)
                for at, t in enumerate(order):
                    got[int(t)] = shuffled[at]
            else:
                got = lib.banded_dp_many(kmer, spec[:, 0], spec[:, 1], spec[:, 2], spec[:, 3], spec[:, 4], spec[:, 5])
            if first[s] is None:
                first[s] = got
                wrong = 0
                for (b0, nx, b1, ny, lo, hi), (y, sy) in zip(spec, got):
                    x, sx = orc.banded_dp(kmer[b0:b0 + nx], kmer[b1:b1 + ny], int(lo), int(hi))
                    wrong += int(sx != sy or not np.array_equal(x, y))
                print("set %d: %d tasks, first run against the oracle: %d differ" % (s, len(spec), wrong), flush=True)
                continue
            for t, ((x, sx), (y, sy)) in enumerate(zip(first[s], got)):
                if sx != sy or not np.array_equal(x, y):
                    bad += 1
                    d = int(np.sum(x != y)) if x.shape == y.shape else -1
                    print("repeat", it, "set", s, "task", t, "spec", spec[t].tolist(), "differs from the first run: scores", sx, sy, "pairs", len(x), len(y), "values that differ", d, flush=True)
    print("library %s: repeats %d x %d + %d tasks, differences between runs %d, %.0f s" % (path or "(the product)", repeats, len(sets[0][1]), len(sets[1][1]), bad, time.time() - t0))


if __name__ == "__main__":
    main()
