"""The one unexplained failure of round 5 (test_kmer_ids_of_k_16, align method 3: aligned pairs of equal number at different positions, once
in 27 runs of the whole -m gpu suite, never alone) looked for in the conditions of a whole-suite run, in ONE process: device memory churned
between the calls (torch tensors of random integers allocated and freed: what a freed buffer holds next is data, not a constant byte), other
aligner calls in between (contexts created and destroyed, their buffers freed), six workers, both option sets of the test, the one-shot and
the device-list forms -- and the ORACLE recomputed every time (a difference between two oracle runs would clear the device).
    python scripts/flake_k16_suite_context.py <repeats>
Prints one line per difference (which side changed, which candidate, how) and a summary line."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
import shasta_amd
from shasta_amd import abi
from oracle import bindings
from tests import config_value_checks as cv, support


def digest(r):
    return (np.asarray(r.status).tobytes(), np.asarray(r.ordinals_toc).tobytes(), np.asarray(r.ordinals).tobytes())


def describe(a, b):
    if not np.array_equal(a.status, b.status):
        return "status differs at %s" % np.nonzero(np.asarray(a.status) != np.asarray(b.status))[0][:5]
    if not np.array_equal(a.ordinals_toc, b.ordinals_toc):
        return "number of aligned markers differs"
    d = np.nonzero(np.asarray(a.ordinals).reshape(-1) != np.asarray(b.ordinals).reshape(-1))[0]
    toc = np.asarray(a.ordinals_toc)
    return "equal numbers, %d values differ, first candidate %d" % (len(d), int(np.searchsorted(toc, d[0] // 2, side="right") - 1))


def main():
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    lib, orc = shasta_amd.load(), bindings.OracleLib()
    dev = torch.device("cuda", 0)
    sets = []
    for k, seed in ((16, 156), (14, 154)):
        toc, kmer, data7 = cv.marker_set(k, 160, 9000, seed=seed, mean_markers=900.0, min_markers=300)
        p = abi.default_lowhash0_params(hashFraction=0.05, **cv.MAY2022_LOWHASH)
        cand = orc.lowhash0(toc, data7, None, p).candidates[:400]
        sets.append((k, toc, data7, cand))
    rng = np.random.default_rng(5)
    bad_device = bad_oracle = calls = 0
    t0 = time.time()
    reference = {}
    for it in range(repeats):
        # Churn: a few gigabytes of random integers on the device, freed again (torch's caching allocator is emptied: the memory goes back).
        junk = [torch.randint(0, 2 ** 31 - 1, (int(rng.integers(1 << 24, 1 << 27)),), dtype=torch.int32, device=dev) for _ in range(3)]
        del junk
        torch.cuda.empty_cache()
        for k, toc, data7, cand in sets:
            for name, kw in (("may2022", cv.MAY2022_ALIGN3), ("fraction", dict(k=k, minAlignedFraction=0.4))):
                o3 = abi.default_align3_options(**kw)
                x = orc.align3_batch(toc, data7, cand, o3, want_ordinals=True, threads=0)
                key = (k, name)
                if key not in reference:
                    reference[key] = x
                elif digest(x) != digest(reference[key]):
                    bad_oracle += 1
                    print("repeat", it, key, "the ORACLE differs from its first run:", describe(reference[key], x), flush=True)
                for form in ("one-shot", "devices (0, 0)"):
                    y = lib.align3_batch(toc, data7, cand, o3, want_ordinals=True) if form == "one-shot" else lib.align3_batch_multi(toc, data7, cand, o3, (0, 0), want_ordinals=True)
                    calls += 1
                    if digest(y) != digest(reference[key]):
                        bad_device += 1
                        print("repeat", it, key, form, "the DEVICE differs:", describe(reference[key], y), flush=True)
            # (something else in between: method 4 on the same reads leaves its own buffers behind)
            o4 = abi.default_align4_options(**cv.MAY2022_ALIGN)
            lib.align4_batch(toc, data7, cand, o4, want_ordinals=True)
    print("repeats %d, method-3 device calls %d, device differences %d, oracle differences %d, %.0f s" % (repeats, calls, bad_device, bad_oracle, time.time() - t0))


if __name__ == "__main__":
    main()
