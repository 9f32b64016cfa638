"""BASELINE.json configs[0] on the box itself: base-level reads with substitution / insertion / deletion and homopolymer-length
errors go through the REFERENCE's own ReadLoader -> MarkerFinder -> LowHash0 -> Align4 control flow (oracle/_ref, the reference's
translation units compiled in place) with conf/Nanopore-Dec2019.conf's values, and through this library from the same stored
reads: marker finding, LowHash0 and the aligner, every output compared.  Nothing is read from a fixture: the reference runs live,
beside the device, on the same bytes.  (The DP inside the reference's aligner is the restated SeqAn call, as everywhere.)"""
import os
import tempfile

import numpy as np

from shasta_amd import abi, synthetic
from tests import support
from tests.config_value_checks import DEC2019_ALIGN, DEC2019_LOWHASH


def plumbing(lib, ref_lib, n_reads=4000, genome_length=1_500_000, mean_length=15000.0, seed=2019, limit=None):
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "reads.fasta")
        synthetic.fasta_reads(path, n_reads, genome_length, mean_length=mean_length, seed=seed, homopolymer=0.03)
        r = ref_lib.reads_and_markers_from_fasta(path, k=10, threads=support.reference_threads(32))        # Kmers.k, probability: the defaults; minReadLength 10000
    read_count = len(r["base_counts"])
    assert read_count >= 0.9 * n_reads
    # Marker finding: the reference's stored reads in, the reference's Markers out.
    with lib.context(0) as ctx:
        toc, data7 = lib.find_markers(r["reads_toc"], r["reads_data"], r["base_counts"], 10, r["is_marker"], context=ctx)
        assert np.array_equal(toc, r["toc"]) and np.array_equal(data7, r["data7"])
        # LowHash0 on the markers that stayed in HBM: m = 4, hashFraction 0.01, 10 iterations, 5/30/5.
        p = abi.default_lowhash0_params(**DEC2019_LOWHASH)
        a = ctx.lowhash0(p)
        b = ref_lib.lowhash0(r["toc"], r["data7"], None, p, threads=support.reference_threads(32))
        support.same_lowhash(a, b)
        cand = b.candidates if limit is None else b.candidates[:limit]
        # computeAlignments, method 4, minAlignedFraction 0.4.
        o = abi.default_align4_options(**DEC2019_ALIGN)
        y = ctx.align4(cand, o, want_ordinals=True)
    x = ref_lib.align4_batch(r["toc"], r["data7"], cand, o, want_ordinals=True, threads=support.reference_threads())      # (never one per core: 2 GiB of arena each)
    assert not (y.status & 0x80).any()
    support.same_align(x, y)
    return read_count, int(r["toc"][-1]), len(b.candidates), int((x.status == abi.SHASTA_ALIGN_STORED).sum())
