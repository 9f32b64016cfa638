"""The stage executable (C++ host layer over the C ABI) on a Data/ directory written by the
reference's containers, against the files and side files the reference's own LowHash0 leaves
behind, and against the oracle's alignments stored through the reference's containers."""
import os
import subprocess

import numpy as np
import pytest

from shasta_amd import abi
from tests import host_support, support

pytestmark = pytest.mark.gpu


def _fields(pairs12):
    a = pairs12.reshape(-1, 12)
    return a[:, 0:9]                      # readIds[2] + isSameStrand; the 3 padding bytes are unspecified in the reference


@pytest.mark.parametrize("devices", ["0", "0,0,0"])
def test_stage_executable_reproduces_the_reference_files(gpu_lib, ref_lib, oracle_lib, tmp_path, monkeypatch, devices):
    # devices = "0,0,0": the C++ host layer runs both seams through the sharded multi-device entry points
    # (SHASTA_MI355X_DEVICES; the one GPU of the box named three times) -- same files, byte for byte.
    monkeypatch.setenv("SHASTA_MI355X_DEVICES", devices)
    toc, kmer, data7 = support.small_marker_set(n_reads=300, genome_markers=20000, seed=91)
    flags = np.zeros(300, np.uint8)
    flags[[2, 250]] = 1
    ref_dir, our_dir = str(tmp_path / "ref"), str(tmp_path / "ours")
    ref_cwd, our_cwd = str(tmp_path / "refcwd"), str(tmp_path / "ourcwd")
    for d in (ref_dir, our_dir, ref_cwd, our_cwd):
        os.makedirs(d)
    ref_lib.write_data_dir(ref_dir, toc, data7, flags)
    ref_lib.write_data_dir(our_dir, toc, data7, flags)
    p = abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30, minFrequency=2)
    ref_lib.lowhash0_files(ref_dir, p, ref_cwd)
    args = ["4", "0.01", "10", "20", "0", "2", "30", "2"]
    out = subprocess.run([host_support.STAGE, "lowhash0", our_dir] + args, cwd=our_cwd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr

    # Seam 1 outputs: same objects in AlignmentCandidates, identical ReadLowHashStatistics file,
    # identical side files, same console lines (except the allocator-dependent "capacity").
    a, size_a = ref_lib.open_vector(os.path.join(ref_dir, "AlignmentCandidates"), 12)
    b, size_b = ref_lib.open_vector(os.path.join(our_dir, "AlignmentCandidates"), 12)
    assert len(a) > 100 and size_a == size_b and np.array_equal(_fields(a), _fields(b))
    assert open(os.path.join(ref_dir, "ReadLowHashStatistics"), "rb").read() == open(os.path.join(our_dir, "ReadLowHashStatistics"), "rb").read()
    for name in ("LowHashBucketHistogram.csv", "ReadLowHashStatistics.csv"):
        assert open(os.path.join(ref_cwd, name)).read() == open(os.path.join(our_cwd, name)).read(), name
    strip = lambda text: [line.split(", capacity")[0] for line in text.strip().splitlines()]
    assert strip(open(os.path.join(ref_cwd, "LowHash0.console")).read()) == strip(out.stdout)

    # Seam 2 on the candidates just written.
    out = subprocess.run([host_support.STAGE, "align", our_dir, "40"], cwd=our_cwd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    cand = abi.make_pairs(a.view("<u4").reshape(-1, 3)[:, 0], a.view("<u4").reshape(-1, 3)[:, 1], a[:, 8])
    al = oracle_lib.align4_batch(toc, data7, cand, abi.default_align4_options(minAlignedMarkerCount=40), want_ordinals=False, threads=0)
    ref_lib.store_alignments(ref_dir, al.alignment_data, al.compressed_toc, al.compressed_data)
    for name in ("CompressedAlignments.toc", "CompressedAlignments.data"):
        assert open(os.path.join(ref_dir, name), "rb").read() == open(os.path.join(our_dir, name), "rb").read(), name
    # AlignmentData: every field equal.  Padding (bytes 9-11, 61-63) is unspecified in the reference
    # (this library writes zeros); numpy does not carry the padding of the oracle's rows either, so
    # the comparison is by field.  isInReadGraph (bit 0 of byte 60) must be cleared (Alignment.hpp:196-209).
    x, size_x = ref_lib.open_vector(os.path.join(ref_dir, "AlignmentData"), 64)
    y, size_y = ref_lib.open_vector(os.path.join(our_dir, "AlignmentData"), 64)
    fx = np.frombuffer(x.tobytes(), dtype=abi.ALIGNMENT_DATA_DTYPE)
    fy = np.frombuffer(y.tobytes(), dtype=abi.ALIGNMENT_DATA_DTYPE)
    assert size_x == size_y and len(fx) == len(fy)
    for field in abi.ALIGNMENT_DATA_DTYPE.names:
        assert np.array_equal(fx[field], fy[field]), field
    assert not (y[:, 60] & 1).any() and not y[:, 9:12].any() and not y[:, 61:64].any()
    assert "Found and stored %d good alignments." % len(al.alignment_data) in out.stdout
    t, _ = ref_lib.open_vector(os.path.join(our_dir, "AlignmentTable.toc"), 4)
    d, _ = ref_lib.open_vector(os.path.join(our_dir, "AlignmentTable.data"), 4)
    toc_expected, data_expected = host_support.alignment_table_expected(300, al.alignment_data)
    assert np.array_equal(t.view("<u4").reshape(-1), toc_expected) and np.array_equal(d.view("<u4").reshape(-1), data_expected)


def test_stage_executable_reports_errors_like_the_reference(tmp_path):
    out = subprocess.run([host_support.STAGE, "lowhash0", str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "Error accessing" in out.stdout


def test_candidate_and_alignment_tables_and_read_graph_selection_on_the_device(gpu_lib, oracle_lib):
    # shasta_mi355x_pair_table / _read_graph_keep against the python restatements of the reference's loops.
    from tests import table_checks
    table_checks.check(gpu_lib, seed=6, read_count=5003, n=200000)
    assert table_checks.table_of_the_last_aligner_call(gpu_lib, oracle_lib) > 300
