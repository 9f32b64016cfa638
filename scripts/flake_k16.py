import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import shasta_amd
from shasta_amd import abi
from oracle import bindings
from tests import config_value_checks as cv, support
lib = shasta_amd.load(); orc = bindings.OracleLib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
toc, kmer, data7 = cv.marker_set(int(os.environ.get('K', '16')), 160, 9000, seed=int(os.environ.get('SEED', '156')), mean_markers=900.0, min_markers=300)
p = abi.default_lowhash0_params(hashFraction=0.05, **cv.MAY2022_LOWHASH)
cand = orc.lowhash0(toc, data7, None, p).candidates[:400]
o3 = abi.default_align3_options(**(cv.MAY2022_ALIGN3 if os.environ.get('O3') != 'b' else dict(k=int(os.environ.get('K', '16')), minAlignedFraction=0.4)))
o4 = abi.default_align4_options(**cv.MAY2022_ALIGN)
x3 = orc.align3_batch(toc, data7, cand, o3, want_ordinals=True, threads=0)
x4 = orc.align4_batch(toc, data7, cand, o4, want_ordinals=True, threads=0)
bad3 = bad4 = 0
for i in range(n):
    y3 = lib.align3_batch(toc, data7, cand, o3, want_ordinals=True)
    y4 = lib.align4_batch(toc, data7, cand, o4, want_ordinals=True)
    ok3 = np.array_equal(x3.status, y3.status) and np.array_equal(x3.ordinals_toc, y3.ordinals_toc) and np.array_equal(x3.ordinals, y3.ordinals)
    ok4 = np.array_equal(x4.status & 0x7f, y4.status & 0x7f) and np.array_equal(x4.ordinals_toc, y4.ordinals_toc) and np.array_equal(x4.ordinals, y4.ordinals)
    if not ok3:
        bad3 += 1
        if np.array_equal(x3.ordinals_toc, y3.ordinals_toc):
            d = np.nonzero(np.asarray(x3.ordinals).reshape(-1) != np.asarray(y3.ordinals).reshape(-1))[0]
            tocs = np.asarray(x3.ordinals_toc)
            print("run", i, "method 3 differs at", len(d), "values; first candidate", int(np.searchsorted(tocs, d[0] // 2, side='right') - 1), "positions", d[:6])
        else:
            print("run", i, "method 3 toc differs")
    if not ok4:
        bad4 += 1; print("run", i, "method 4 differs")
print("runs", n, "method 3 bad", bad3, "method 4 bad", bad4)
