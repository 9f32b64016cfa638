#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: what one steady-state step asks of the HIP runtime -- launches, synchronisations, copies, allocations -- counted by
the emulated runtime (tests/emu/hip_emu.cpp: hipemu_api_counts) while the unmodified host code of shasta_amd/csrc drives it.  The host
code does not know it is emulated: the counts per batch are those of a run on the device (their cost is not: that is the device's).
    python scripts/emu_api_counts.py [reads] [workers] [path of another emulated build's .so]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    numbers = [a for a in sys.argv[1:] if not a.endswith(".so")]
    path = next((a for a in sys.argv[1:] if a.endswith(".so")), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "emu", "_build", "libshasta_mi355x_emu.so"))
    reads = int(numbers[0]) if numbers else 400
    if len(numbers) > 1:
        os.environ["SHASTA_MI355X_ALIGN_WORKERS"] = numbers[1]
    from shasta_amd import abi, lib as L
    from tests import support
    emu = L.Library(path)
    raw = emu.lib
    raw.hipemu_api_count_names.restype = C.c_char_p
    names = raw.hipemu_api_count_names().decode().split()

    def counts():
        out = (C.c_uint64 * 32)()
        n = raw.hipemu_api_counts(out, 32)
        return [int(out[k]) for k in range(n)]

    toc, kmer, data7 = support.small_marker_set(n_reads=reads, genome_markers=30 * reads, seed=5)
    p = abi.default_lowhash0_params(minBucketSize=3, maxBucketSize=30, minFrequency=2)
    o = abi.default_align4_options(minAlignedMarkerCount=40)
    with emu.context(0) as ctx:
        ctx.set_kmer_ids(toc, kmer)
        rows = []
        for step in range(3):
            a = counts()
            lh = ctx.lowhash0(p)
            b = counts()
            al = ctx.align4(lh.candidates, o, want_ordinals=False, borrow=True)
            c = counts()
            ctx.alignment_table(copy=False)
            d = counts()
            rows = [("LowHash0", a, b), ("Align4 (%d candidates, one batch)" % len(lh.candidates), b, c), ("alignment table", c, d)]
        print("third step of LowHash0 + Align4 + alignment table on one context, %d reads:" % reads)
        for what, x, y in rows:
            print("  %-40s" % what, ", ".join("%s %d" % (n, v - u) for n, u, v in zip(names, x, y) if v - u))


if __name__ == "__main__":
    main()
