import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import shasta_amd
from shasta_amd import abi
from oracle import bindings
from tests import support
lib = shasta_amd.load(); orc = bindings.OracleLib()
g = support.Golden(sys.argv[1] if len(sys.argv) > 1 else "tiny.npz")
o = abi.default_align4_options()
cand = g.candidates(0)
a = orc.align4_batch(g.toc, g.data7, cand, o, want_ordinals=True, threads=0)
b = lib.align4_batch(g.toc, g.data7, cand, o, want_ordinals=True)
print("status oracle", np.bincount(a.status & 0x7f, minlength=4), "gpu", np.bincount(b.status & 0x7f, minlength=4))
bad = np.nonzero((a.status & 0x7f) != (b.status & 0x7f))[0]
print("mismatching status:", len(bad), bad[:20])
toc = g.toc.astype(np.int64)
for i in bad[:10]:
    c = cand[i]
    o0 = 2 * int(c["readId0"]); o1 = 2 * int(c["readId1"]) + (0 if c["isSameStrand"] else 1)
    print(i, c, "nx", toc[o0+1]-toc[o0], "ny", toc[o1+1]-toc[o1], "oracle", a.status[i], len(a.ordinals_of(i)), "gpu", b.status[i], len(b.ordinals_of(i)))
same = np.nonzero((a.status & 0x7f) == (b.status & 0x7f))[0]
nd = 0
for i in same:
    if not np.array_equal(a.ordinals_of(i), b.ordinals_of(i)):
        nd += 1
        if nd < 5:
            x, y = a.ordinals_of(i), b.ordinals_of(i)
            print("ordinals differ", i, len(x), len(y))
print("ordinal mismatches among same-status:", nd)
