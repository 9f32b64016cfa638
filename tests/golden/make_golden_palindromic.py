"""Golden fixture for palindromic-read flagging, made by RUNNING THE REFERENCE'S OWN
alignment method 0 (/root/reference/src/AlignmentGraph.cpp compiled in place into oracle/_ref; the caller
loop of Assembler::flagPalindromicReadsThreadFunction is restated in oracle/ref_build/ref_palindromic.cpp)
on the real reads of tiny.npz and on tests/palindromic_checks.read_set.  Run from the repo root in the
build container:

    python tests/golden/make_golden_palindromic.py

Writes palindromic.npz (loaded by tests/palindromic_checks.golden).
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bindings  # noqa: E402
from tests import palindromic_checks as pc, support  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ref = bindings.RefLib()
    g = support.Golden("tiny.npz")
    toc, kmer, data7, kinds = pc.read_set(**pc.GOLDEN_READ_SET)
    out = {"hairpins_input_md5": hashlib.md5(kmer.tobytes()).hexdigest(), "hairpins_kinds": np.asarray(kinds)}
    for name, (t, d) in (("tiny", (g.toc, g.data7)), ("hairpins", (toc, data7))):
        for i, kw in enumerate(pc.PARAMETER_SETS):
            flags, aligned, near, digests = ref.flag_palindromic_reads(t, d, threads=4, **kw)
            one = ref.flag_palindromic_reads(t, d, threads=1, **kw)
            assert all(np.array_equal(x, y) for x, y in zip((flags, aligned, near, digests), one)), "thread count changed the answer"
            out["%s_%d_flags" % (name, i)] = flags
            out["%s_%d_aligned" % (name, i)] = aligned
            out["%s_%d_near" % (name, i)] = near
            out["%s_%d_digests" % (name, i)] = digests
            print(name, i, "flagged", int(flags.sum()), "of", len(flags), "aligned markers", int(aligned.sum()))
    np.savez_compressed(os.path.join(HERE, "palindromic.npz"), **out)


if __name__ == "__main__":
    main()
