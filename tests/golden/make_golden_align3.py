"""Golden fixtures for align method 3, made by RUNNING THE REFERENCE'S OWN
Assembler::alignOrientedReads3 (/root/reference/src/AssemblerAlign3.cpp compiled in place into
oracle/_ref, see oracle/ref_build/ref_align3.cpp) on the marker sets of tiny.npz / synth.npz and
their first LowHash0 candidate list.  Run from the repo root in the build container:

    python tests/golden/make_golden_align3.py

Writes tiny_align3.npz and synth_align3.npz (loaded by tests/support.py: check_align3).
NB the DP behind both steps is the restated one (SeqAn is absent): parity UNPINNED there.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bindings  # noqa: E402
from shasta_amd import abi  # noqa: E402
from tests import support  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ref = bindings.RefLib()
    for name in ("tiny", "synth"):
        g = support.Golden(name + ".npz")
        cand = g.candidates(0)
        out = {}
        for i, kw in enumerate(support.ALIGN3_OPTION_SETS):
            o = abi.default_align3_options(**kw)
            a = ref.align3_batch(g.toc, g.data7, cand, o, want_ordinals=True, threads=4)
            out["m3_%d_status" % i] = a.status
            out["m3_%d_info" % i] = a.info_table()
            out["m3_%d_compressed_toc" % i] = a.compressed_toc
            out["m3_%d_compressed_data" % i] = a.compressed_data
            out["m3_%d_marker_count" % i] = np.diff(a.ordinals_toc.astype(np.int64))
            h = hashlib.md5(a.ordinals.tobytes()).hexdigest()
            out["m3_%d_ordinals_md5" % i] = np.frombuffer(h.encode(), dtype=np.uint8)
            print(name, i, kw, "stored", len(a.alignment_data), "status", np.bincount(a.status).tolist())
        path = os.path.join(HERE, name + "_align3.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
