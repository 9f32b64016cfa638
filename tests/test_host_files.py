"""The C++ host layer (shasta_amd/host/): Shasta's Data/ file formats against the reference's own
MemoryMapped containers (oracle/_ref), and computeAlignmentTable against its specification."""
import os

import numpy as np
import pytest

from shasta_amd import abi
from tests import host_support, support


# The table steps (candidate table, alignment table, read graph selection) run on the device: on the wave64 emulator
# here, on the MI355X in the -m gpu run.
@pytest.fixture(scope="module", params=["emulated", pytest.param("mi355x", marks=pytest.mark.gpu)])
def shim(request):
    if request.param == "emulated":
        host_support.emulated_build()
        return host_support.HostShim(host_support.EMU_SHIM)
    if not os.path.exists(host_support.SHIM):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(host_support.ROOT, "shasta_amd", "csrc"), "host"])
    return host_support.HostShim()


def test_files_written_by_the_host_layer_open_in_the_reference(shim, ref_lib, tmp_path):
    toc, kmer, data7 = support.small_marker_set(n_reads=60, genome_markers=5000, seed=81)
    flags = np.zeros(60, np.uint8)
    flags[[5, 9]] = 1
    d = str(tmp_path)
    shim.write_data_dir(d, toc, data7, flags)
    # The reference's accessExistingReadOnly checks magic number, file size and object size.
    m_toc, size = ref_lib.open_vector(os.path.join(d, "Markers.toc"), 8)
    assert size == os.path.getsize(os.path.join(d, "Markers.toc")) and size % 4096 == 0
    assert np.array_equal(m_toc.view("<u8").reshape(-1), np.asarray(toc, np.uint64))
    m_data, size = ref_lib.open_vector(os.path.join(d, "Markers.data"), 7)
    assert 0 <= os.path.getsize(os.path.join(d, "Markers.data")) - size < 7     # capacity is a whole number of objects
    assert np.array_equal(m_data.reshape(-1), data7)
    f, _ = ref_lib.open_vector(os.path.join(d, "ReadFlags"), 1)
    assert np.array_equal(f.reshape(-1), flags)
    with pytest.raises(RuntimeError, match="unexpected object size"):
        ref_lib.open_vector(os.path.join(d, "Markers.data"), 8)


def test_files_written_by_the_reference_open_in_the_host_layer_and_are_byte_identical(shim, ref_lib, tmp_path):
    toc, kmer, data7 = support.small_marker_set(n_reads=60, genome_markers=5000, seed=82)
    a, b = str(tmp_path / "ref"), str(tmp_path / "host")
    os.makedirs(a); os.makedirs(b)
    ref_lib.write_data_dir(a, toc, data7, None)
    shim.write_data_dir(b, toc, data7, None)
    for name, size in (("Markers.toc", 8), ("Markers.data", 7), ("ReadFlags", 1)):
        x, _ = shim.open_vector(os.path.join(a, name), size)
        y, _ = ref_lib.open_vector(os.path.join(a, name), size)
        assert np.array_equal(x, y)
        # Same header and same bytes as the reference's own file, after unreserve().
        assert open(os.path.join(a, name), "rb").read() == open(os.path.join(b, name), "rb").read(), name
    with pytest.raises(RuntimeError, match="unexpected object size"):
        shim.open_vector(os.path.join(a, "Markers.toc"), 4)
    with open(os.path.join(a, "ReadFlags"), "r+b") as f:
        f.seek(56); f.write(b"\0" * 8)                       # clobber the magic number
    with pytest.raises(RuntimeError, match="unexpected magic number"):
        shim.open_vector(os.path.join(a, "ReadFlags"), 1)


def test_stored_alignments_and_alignment_table(shim, ref_lib, oracle_lib, tmp_path):
    toc, kmer, data7 = support.small_marker_set(n_reads=150, genome_markers=9000, seed=83)
    p = abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30)
    cand = oracle_lib.lowhash0(toc, data7, None, p).candidates
    al = oracle_lib.align4_batch(toc, data7, cand, abi.default_align4_options(minAlignedMarkerCount=40), want_ordinals=False, threads=0)
    assert len(al.alignment_data) > 50
    a, b = str(tmp_path / "ref"), str(tmp_path / "host")
    os.makedirs(a); os.makedirs(b)
    ref_lib.store_alignments(a, al.alignment_data, al.compressed_toc, al.compressed_data)
    shim.store_alignments(b, al.alignment_data, al.compressed_toc, al.compressed_data)
    for name in ("AlignmentData", "CompressedAlignments.toc", "CompressedAlignments.data"):
        assert open(os.path.join(a, name), "rb").read() == open(os.path.join(b, name), "rb").read(), name
    shim.compute_alignment_table(b, 150)
    t, _ = ref_lib.open_vector(os.path.join(b, "AlignmentTable.toc"), 4)
    dta, _ = ref_lib.open_vector(os.path.join(b, "AlignmentTable.data"), 4)
    toc_expected, data_expected = host_support.alignment_table_expected(150, al.alignment_data)
    assert np.array_equal(t.view("<u4").reshape(-1), toc_expected)
    assert np.array_equal(dta.view("<u4").reshape(-1), data_expected)


def test_candidate_table(shim, ref_lib, oracle_lib, tmp_path):
    toc, kmer, data7 = support.small_marker_set(n_reads=150, genome_markers=9000, seed=84)
    cand = oracle_lib.lowhash0(toc, data7, None, abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30)).candidates
    assert len(cand) > 200
    d = str(tmp_path)
    shim.store_candidates(d, cand)
    stored, _ = ref_lib.open_vector(os.path.join(d, "AlignmentCandidates"), 12)      # opens in the reference
    assert np.array_equal(stored.view("<u4").reshape(-1, 3)[:, 0], cand["readId0"])
    shim.compute_candidate_table(d, 150)
    t, _ = ref_lib.open_vector(os.path.join(d, "CandidateTable.toc"), 8)
    dta, _ = ref_lib.open_vector(os.path.join(d, "CandidateTable.data"), 8)
    toc_expected, data_expected = host_support.alignment_table_expected(150, cand, np.uint64)
    assert np.array_equal(t.view("<u8").reshape(-1), toc_expected)
    assert np.array_equal(dta.view("<u8").reshape(-1), data_expected)


@pytest.mark.parametrize("max_alignment_count", [2, 6])
def test_read_graph_selection(shim, ref_lib, oracle_lib, tmp_path, max_alignment_count):
    toc, kmer, data7 = support.small_marker_set(n_reads=150, genome_markers=9000, seed=87)
    cand = oracle_lib.lowhash0(toc, data7, None, abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30)).candidates
    al = oracle_lib.align4_batch(toc, data7, cand, abi.default_align4_options(minAlignedMarkerCount=40), want_ordinals=False, threads=0)
    rows = al.alignment_data
    assert len(rows) > 150
    d = str(tmp_path)
    shim.store_alignments(d, rows, al.compressed_toc, al.compressed_data)
    shim.compute_alignment_table(d, 150)
    kept = shim.create_read_graph(d, max_alignment_count)
    keep, edges, ctoc, cdata = host_support.read_graph_expected(150, rows, max_alignment_count)
    assert kept == int(keep.sum()) and 0 < kept <= len(rows) and (max_alignment_count == 6 or kept < len(rows))
    # isInReadGraph is set in place in AlignmentData (bit 0 of byte 60); nothing else changes.
    after, _ = ref_lib.open_vector(os.path.join(d, "AlignmentData"), 64)
    assert np.array_equal((after[:, 60] & 1).astype(bool), keep)
    got = np.frombuffer(after.tobytes(), dtype=abi.ALIGNMENT_DATA_DTYPE)
    for field in abi.ALIGNMENT_DATA_DTYPE.names:
        assert np.array_equal(got[field], rows[field]), field
    e, _ = ref_lib.open_vector(os.path.join(d, "ReadGraphEdges"), 16)
    assert len(e) == len(edges)
    o = e[:, 0:8].copy().view("<u4").reshape(-1, 2)
    ids = e[:, 8:16].copy().view("<u8").reshape(-1)
    assert np.array_equal(o, np.asarray([(a, b) for a, b, _ in edges], np.uint32).reshape(-1, 2))
    assert np.array_equal(ids, np.asarray([i for _, _, i in edges], np.uint64))          # flag bits 62, 63 are zero
    assert np.all(o[:, 0] < o[:, 1])
    t, _ = ref_lib.open_vector(os.path.join(d, "ReadGraphConnectivity.toc"), 4)
    c, _ = ref_lib.open_vector(os.path.join(d, "ReadGraphConnectivity.data"), 4)
    assert np.array_equal(t.view("<u4").reshape(-1), ctoc) and np.array_equal(c.view("<u4").reshape(-1), cdata)
