"""Generates the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF
(oracle/_ref/libshasta_ref.so = /root/reference/src compiled in place; see
oracle/ref_build/).  Run from the repo root in the build container:

    python tests/golden/make_golden.py

Fixtures (all .npz, loaded by tests/support.py):
  tiny.npz    markers of /root/reference/tests/TinyTest.fasta.gz via the reference's
              ReadLoader + MarkerFinder (k=10, probability 0.1, seed 231, minReadLength 10000),
              LowHash0 outputs for three parameter sets, Align4 outputs (default options).
  synth.npz   the same for a seeded synthetic read set (shasta_amd.synthetic.fasta_reads).
SURVEY.md section 8c digests are asserted for tiny.npz before anything is written.
NB the banded DP inside Align4 is the restated one (SeqAn is absent): parity UNPINNED there.
"""
import gzip
import hashlib
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bindings  # noqa: E402
from shasta_amd import abi, synthetic  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

LOWHASH_PARAM_SETS = [
    dict(),                                                          # src/AssemblerOptions.cpp defaults
    dict(m=5, minBucketSize=2, maxBucketSize=5, minFrequency=3),     # odd m exercises the hash tail
    dict(m=3, hashFraction=0.05, minHashIterationCount=0, alignmentCandidatesPerRead=12.0),
]
ALIGN_OPTION_SETS = [
    dict(),
    dict(minAlignedMarkerCount=10, minAlignedFraction=0.1, maxSkip=100, maxDrift=100, maxTrim=100,
         suppressContainments=1),                                     # Nanopore-May2022 [Align] values
]


def run(ref, toc, data7, out):
    flags = np.zeros((len(toc) - 1) // 2, np.uint8)
    flags[3] = 1                                                     # one palindromic read in set 1
    for i, kw in enumerate(LOWHASH_PARAM_SETS):
        p = abi.default_lowhash0_params(**kw)
        r = ref.lowhash0(toc, data7, flags if i == 1 else None, p, threads=3)
        out["lh%d_candidates" % i] = r.candidate_tuples().astype(np.uint32)
        out["lh%d_statistics" % i] = r.statistics
        out["lh%d_high" % i] = r.high_frequency
        out["lh%d_total" % i] = r.total
        out["lh%d_histogram" % i] = r.histogram
        out["lh%d_log2" % i] = np.array([r.log2_bucket_count])
        if i == 0:
            cand = r.candidates
            first = r
    out["flags1"] = flags
    for i, kw in enumerate(ALIGN_OPTION_SETS):
        o = abi.default_align4_options(**kw)
        a = ref.align4_batch(toc, data7, cand, o, want_ordinals=True)
        out["al%d_status" % i] = a.status
        out["al%d_info" % i] = a.info_table()
        out["al%d_compressed_toc" % i] = a.compressed_toc
        out["al%d_compressed_data" % i] = a.compressed_data
        out["al%d_ordinals_toc" % i] = a.ordinals_toc
        out["al%d_marker_count" % i] = np.diff(a.ordinals_toc.astype(np.int64))
        h = hashlib.md5(a.ordinals.tobytes()).hexdigest()
        out["al%d_ordinals_md5" % i] = np.frombuffer(h.encode(), dtype=np.uint8)
    return first


def save(name, toc, data7, out):
    d = np.ascontiguousarray(data7, np.uint8).reshape(-1, 7)
    out["toc"] = np.asarray(toc, np.uint64)
    out["kmer_ids"] = np.ascontiguousarray(d[:, 0:4]).view("<u4").reshape(-1)
    pos = np.zeros((len(d), 4), np.uint8)
    pos[:, 0:3] = d[:, 4:7]
    out["positions"] = pos.view("<u4").reshape(-1)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")


def main():
    ref = bindings.RefLib()
    ref.test_alignment_compression()
    with tempfile.TemporaryDirectory() as tmp:
        fasta = os.path.join(tmp, "TinyTest.fasta")
        with gzip.open("/root/reference/tests/TinyTest.fasta.gz", "rb") as f, open(fasta, "wb") as g:
            shutil.copyfileobj(f, g)
        assert hashlib.md5(open(fasta, "rb").read()).hexdigest() == "ec9795c206fe5fc49c3f272809595326"
        toc, data7 = ref.markers_from_fasta(fasta)
        out = {}
        r = run(ref, toc, data7, out)
        # SURVEY.md 8c golden digests.
        txt = "".join("%d %d %d\n" % tuple(c) for c in r.candidate_tuples()) + \
            "".join("S %d %d %d\n" % tuple(s) for s in r.statistics)
        assert hashlib.md5(txt.encode()).hexdigest() == "aec61b27e701056f2899e260b8ed26dd"
        assert hashlib.md5(r.histogram_csv).hexdigest() == "ee28cb15dd9b1b55229702066e6b0e05"
        assert list(r.high_frequency) == [127, 161, 165, 168, 170, 175, 182, 182, 184, 186]
        save("tiny.npz", toc, data7, out)

        fasta = os.path.join(tmp, "synth.fasta")
        synthetic.fasta_reads(fasta, n_reads=70, genome_length=60000, mean_length=13000.0, seed=777)
        toc, data7 = ref.markers_from_fasta(fasta)
        out = {}
        run(ref, toc, data7, out)
        save("synth.npz", toc, data7, out)


if __name__ == "__main__":
    main()
