import sys
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from shasta_amd import abi
from oracle import bindings
from tests import config_value_checks as cv
orc = bindings.OracleLib()
toc, kmer, data7 = cv.marker_set(16, 160, 9000, seed=156, mean_markers=900.0, min_markers=300)
p = abi.default_lowhash0_params(hashFraction=0.05, **cv.MAY2022_LOWHASH)
cand = orc.lowhash0(toc, data7, None, p).candidates[:400]
for name, o3 in (("a", abi.default_align3_options(**cv.MAY2022_ALIGN3)), ("b", abi.default_align3_options(k=16, minAlignedFraction=0.4))):
    x = orc.align3_batch(toc, data7, cand, o3, want_ordinals=True, threads=1)
    bad = 0
    for i in range(int(sys.argv[1])):
        y = orc.align3_batch(toc, data7, cand, o3, want_ordinals=True, threads=0)
        if not (np.array_equal(x.status, y.status) and np.array_equal(x.ordinals_toc, y.ordinals_toc) and np.array_equal(x.ordinals, y.ordinals)):
            bad += 1
    print(name, "oracle runs", sys.argv[1], "different from the one-thread run:", bad)
