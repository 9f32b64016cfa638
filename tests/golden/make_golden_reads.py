"""Golden input of marker finding: what the reference's MarkerFinder READ to produce the markers of
tiny.npz -- the reads of /root/reference/tests/TinyTest.fasta.gz as the reference's ReadLoader stores
them (RLE, two bit planes per 64 bases) and the isMarker flag of every k-mer id (k=10, probability
0.1, seed 231).  Run from the repo root in the build container:

    python tests/golden/make_golden_reads.py

Writes tiny_reads.npz; asserts that the markers the same reference run wrote equal those of tiny.npz.
"""
import gzip
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bindings  # noqa: E402
from tests import support  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ref = bindings.RefLib()
    with tempfile.TemporaryDirectory() as tmp:
        fasta = os.path.join(tmp, "TinyTest.fasta")
        with gzip.open("/root/reference/tests/TinyTest.fasta.gz", "rb") as f, open(fasta, "wb") as g:
            shutil.copyfileobj(f, g)
        z = ref.reads_and_markers_from_fasta(fasta)
    g = support.Golden("tiny.npz")
    assert np.array_equal(z["toc"], g.toc) and np.array_equal(z["data7"], g.data7)
    path = os.path.join(HERE, "tiny_reads.npz")
    np.savez_compressed(path, reads_toc=z["reads_toc"], reads_data=z["reads_data"], base_counts=z["base_counts"],
                        is_marker_bits=np.packbits(z["is_marker"], bitorder="little"))
    print(path, os.path.getsize(path), "bytes;", len(z["base_counts"]), "reads,", int(z["base_counts"].sum()), "bases")


if __name__ == "__main__":
    main()
