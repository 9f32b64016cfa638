"""Round 5's one unexplained failure, reproduced in round 6 (scripts/flake_k16_suite_context.py, profiles/r06_flake2_context_3000.log:
1 of 24 000 method-3 calls, in the DEVICE-LIST form with one device listed twice, 8 aligned pairs of one candidate at other positions):
that form only, with the device listed FLAKE_DEVICES times (default 4: as many contexts working side by side on the one GPU), against
results of the oracle computed once.  FLAKE_LIBRARY=<path of a libshasta_mi355x.so>: another build (the A/B with vector stores of the
trace: make -C shasta_amd/csrc OUT=../_build_vector_stores EXTRA=-DSHASTA_TRACE_VECTOR_STORES=1).
    python scripts/flake_multi_form.py <repeats>"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch   # noqa: F401  (first, as in the suite: the torch wheel's HIP runtime is the one the process uses)
import shasta_amd
from shasta_amd import abi, lib as libmod
from oracle import bindings
from tests import config_value_checks as cv
from scripts.flake_k16_suite_context import digest, describe


def main():
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    path = os.environ.get("FLAKE_LIBRARY")
    lib = libmod.Library(path) if path else shasta_amd.load()
    devices = tuple([0] * int(os.environ.get("FLAKE_DEVICES", "4")))
    method4_too = os.environ.get("FLAKE_METHOD4") == "1"
    orc = bindings.OracleLib()
    jobs = []
    for k, seed in ((16, 156), (14, 154)):
        toc, kmer, data7 = cv.marker_set(k, 160, 9000, seed=seed, mean_markers=900.0, min_markers=300)
        p = abi.default_lowhash0_params(hashFraction=0.05, **cv.MAY2022_LOWHASH)
        cand = orc.lowhash0(toc, data7, None, p).candidates[:400]
        for name, kw in (("may2022", cv.MAY2022_ALIGN3), ("fraction", dict(k=k, minAlignedFraction=0.4))):
            o3 = abi.default_align3_options(**kw)
            jobs.append(((k, name, 3), toc, data7, cand, o3, orc.align3_batch(toc, data7, cand, o3, want_ordinals=True, threads=0)))
        if method4_too:
            # (SHASTA_MI355X_SPARSE_DP=0 in the environment: method 4's tasks through the same dense kernels)
            o4 = abi.default_align4_options(**cv.MAY2022_ALIGN)
            jobs.append(((k, "may2022", 4), toc, data7, cand, o4, orc.align4_batch(toc, data7, cand, o4, want_ordinals=True, threads=0)))
    bad = calls = 0
    t0 = time.time()
    for it in range(repeats):
        for key, toc, data7, cand, o, want in jobs:
            got = (lib.align3_batch_multi if key[2] == 3 else lib.align4_batch_multi)(toc, data7, cand, o, devices, want_ordinals=True)
            calls += 1
            if digest(got) != digest(want):
                bad += 1
                print("repeat", it, key, "the DEVICE differs:", describe(want, got), flush=True)
                # Everything about the first candidate that differs: both lists of aligned pairs, the rows, and what the same call gives when
                # it is made again at once (does the difference stay?).
                try:
                    wt, gt = np.asarray(want.ordinals_toc).astype(np.int64), np.asarray(got.ordinals_toc).astype(np.int64)
                    if np.array_equal(wt, gt):
                        d = np.nonzero(np.asarray(want.ordinals).reshape(-1) != np.asarray(got.ordinals).reshape(-1))[0]
                        c = int(np.searchsorted(wt, d[0] // 2, side="right") - 1)
                        again = (lib.align3_batch_multi if key[2] == 3 else lib.align4_batch_multi)(toc, data7, cand, o, devices, want_ordinals=True)
                        w, g, a = want.ordinals_of(c), got.ordinals_of(c), again.ordinals_of(c)
                        rows = np.nonzero((w != g).any(axis=1))[0]
                        print("   candidate", c, tuple(int(x) for x in (cand["readId0"][c], cand["readId1"][c], cand["isSameStrand"][c])), "pairs", len(w),
                              "rows that differ", rows[:12].tolist(), "... of", len(rows), "; the call made again equals the oracle:", bool(np.array_equal(a, w)),
                              "equals the first answer:", bool(np.array_equal(a, g)), flush=True)
                        lo, hi = max(0, int(rows[0]) - 3), min(len(w), int(rows[-1]) + 4)
                        print("   oracle", w[lo:hi].tolist(), flush=True)
                        print("   device", g[lo:hi].tolist(), flush=True)
                        np.savez(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06_flake_event_%d_%d.npz" % (it, bad)), oracle=w, device=g, again=a,
                                 candidate=np.array([c, key[0], key[2]]), name=np.array([key[1]]))
                except Exception as e:          # noqa: BLE001
                    print("   (could not describe it further: %s)" % e, flush=True)
    print("library %s, %d contexts on device 0: repeats %d, calls %d, differences %d, %.0f s" % (path or "(the product)", len(devices), repeats, calls, bad, time.time() - t0))


if __name__ == "__main__":
    main()
