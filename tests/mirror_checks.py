"""The stage functions through shasta_amd.assembler (the reference's Python stage surface) on a Data/
directory, outputs against the oracle; shared by the GPU test and its emulated pre-flight."""
import os

import numpy as np

import shasta_amd.assembler as shasta
from shasta_amd import abi
from tests import host_support, support


def stages_on_a_data_directory(oracle_lib, tmp_path, monkeypatch, host_library, align_method):
    toc, kmer, data7 = support.small_marker_set(n_reads=120, genome_markers=9000, seed=86 + align_method)
    d = str(tmp_path / "Data")
    os.makedirs(d)
    shim = host_support.HostShim()
    shim.write_data_dir(d, toc, data7, None)
    shim.write_kmers(d, 10)
    monkeypatch.chdir(tmp_path)                                   # the CSV side files go to the run directory
    a = shasta.Assembler(hostLibrary=host_library)                # default prefix "Data/", as in the reference
    a.accessKmers(); a.accessMarkers()
    a.findAlignmentCandidatesLowHash0(m=4, hashFraction=0.01, minHashIterationCount=10, alignmentCandidatesPerRead=20.0,
                                      minBucketSize=2, maxBucketSize=30, minFrequency=2)
    ref = oracle_lib.lowhash0(toc, data7, None, abi.default_lowhash0_params(minBucketSize=2, maxBucketSize=30, minFrequency=2))
    stored, _ = shim.open_vector(os.path.join(d, "AlignmentCandidates"), 12)
    assert np.array_equal(stored.view("<u4").reshape(-1, 3)[:, :2], ref.candidate_tuples()[:, :2])
    assert np.array_equal(stored[:, 8], ref.candidate_tuples()[:, 2].astype(np.uint8))
    a.computeCandidateTable()
    a.accessAlignmentCandidates()
    o = shasta.AlignOptions()
    o.alignMethod = align_method
    o.minAlignedMarkerCount = 40
    a.computeAlignments(o, 0)
    if align_method == 3:
        al = oracle_lib.align3_batch(toc, data7, ref.candidates, abi.default_align3_options(minAlignedMarkerCount=40),
                                     want_ordinals=False, threads=0)
    else:
        al = oracle_lib.align4_batch(toc, data7, ref.candidates, abi.default_align4_options(minAlignedMarkerCount=40),
                                     want_ordinals=False, threads=0)
    rows, _ = shim.open_vector(os.path.join(d, "AlignmentData"), 64)
    got = np.frombuffer(rows.tobytes(), dtype=abi.ALIGNMENT_DATA_DTYPE)
    assert len(got) == len(al.alignment_data) > 20
    for field in abi.ALIGNMENT_DATA_DTYPE.names:
        assert np.array_equal(got[field], al.alignment_data[field]), field
    blob, _ = shim.open_vector(os.path.join(d, "CompressedAlignments.data"), 1)
    assert np.array_equal(blob.reshape(-1), al.compressed_data)
    a.accessAlignmentData()
    a.createReadGraph(6, 30)
    assert os.path.exists(os.path.join(d, "ReadGraphEdges"))


def find_markers_on_a_data_directory(tmp_path, host_library):
    """Reads-Bases / Reads-BaseCount / Kmers in, Markers.{toc,data} out: the files the reference's
    findMarkers wrote for the same reads (tiny.npz)."""
    from tests import marker_checks
    rt, rd, bc, im = marker_checks.tiny_reads()
    g = support.Golden("tiny.npz")
    d = str(tmp_path / "Data")
    os.makedirs(d)
    shim = host_support.HostShim()
    shim.write_reads(d, rt, rd, bc)
    shim.write_kmers(d, 10, im)
    a = shasta.Assembler(d, hostLibrary=host_library)
    a.findMarkers()
    toc, _ = shim.open_vector(os.path.join(d, "Markers.toc"), 8)
    data, _ = shim.open_vector(os.path.join(d, "Markers.data"), 7)
    assert np.array_equal(toc.view("<u8").reshape(-1), g.toc)
    assert np.array_equal(data.reshape(-1), g.data7)



def whole_chain_on_the_tiny_reads(oracle_lib, tmp_path, monkeypatch, host_library):
    """Reads -> markers -> palindromic flags -> LowHash0 -> candidate table -> Align4 -> read graph through the Python
    mirror on one Data/ directory, every stage reading the files the previous one wrote; against what the
    reference produced from the same 20 real reads (tiny.npz, palindromic.npz: made by the reference's own code)."""
    from tests import marker_checks, palindromic_checks as pc
    rt, rd, bc, im = marker_checks.tiny_reads()
    g = support.Golden("tiny.npz")
    read_count = len(bc)
    d = str(tmp_path / "Data")
    os.makedirs(d)
    shim = host_support.HostShim()
    shim.write_reads(d, rt, rd, bc)
    shim.write_kmers(d, 10, im)
    shim.write_read_flags(d, read_count, np.ones(read_count, np.uint8))     # stale flags: the stage must reset them
    monkeypatch.chdir(tmp_path)
    a = shasta.Assembler(hostLibrary=host_library)
    a.accessKmers()
    a.findMarkers()
    toc, _ = shim.open_vector(os.path.join(d, "Markers.toc"), 8)
    assert np.array_equal(toc.view("<u8").reshape(-1), g.toc)
    a.accessMarkers()
    counts = a.flagPalindromicReads(100, 100, 10, 0.1, 0.1, 100)
    z, _ = pc.golden()
    flags, _ = shim.open_vector(os.path.join(d, "ReadFlags"), 1)
    assert np.array_equal(flags.reshape(-1) & 1, z["tiny_0_flags"]) and counts[0] == read_count and counts[2] == 0
    a.findAlignmentCandidatesLowHash0(m=4, hashFraction=0.01, minHashIterationCount=10, alignmentCandidatesPerRead=20.0,
                                      minBucketSize=0, maxBucketSize=10, minFrequency=2)
    stored, _ = shim.open_vector(os.path.join(d, "AlignmentCandidates"), 12)
    expected = g.z["lh0_candidates"]
    assert np.array_equal(stored.view("<u4").reshape(-1, 3)[:, :2], expected[:, :2])
    assert np.array_equal(stored[:, 8], expected[:, 2].astype(np.uint8))
    a.computeCandidateTable()
    a.accessAlignmentCandidates()
    a.computeAlignments(shasta.AlignOptions(), 0)
    al = oracle_lib.align4_batch(g.toc, g.data7, g.candidates(0), abi.default_align4_options(), want_ordinals=False, threads=0)
    rows, _ = shim.open_vector(os.path.join(d, "AlignmentData"), 64)
    got = np.frombuffer(rows.tobytes(), dtype=abi.ALIGNMENT_DATA_DTYPE)
    assert len(got) == len(al.alignment_data) > 50
    for field in abi.ALIGNMENT_DATA_DTYPE.names:
        assert np.array_equal(got[field], al.alignment_data[field]), field
    a.accessAlignmentData()
    a.createReadGraph(6, 30)
    assert os.path.exists(os.path.join(d, "ReadGraphEdges"))


def stage_scripts_in_a_run_directory(oracle_lib, tmp_path, host_library):
    """The stage scripts of scripts/ (the reference's own stage scripts with `import shasta_amd.assembler as shasta`)
    run one after the other as processes in a run directory, on the 20 real reads; outputs against the reference's."""
    import subprocess
    import sys
    from tests import marker_checks
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rt, rd, bc, im = marker_checks.tiny_reads()
    g = support.Golden("tiny.npz")
    d = str(tmp_path / "Data")
    os.makedirs(d)
    shim = host_support.HostShim()
    shim.write_reads(d, rt, rd, bc)
    shim.write_kmers(d, 10, im)
    shim.write_read_flags(d, len(bc), None)
    env = dict(os.environ, SHASTA_MI355X_HOST_LIBRARY=host_library)
    for script, args in (("FindMarkers.py", []), ("FlagPalindromicReads.py", ["deltaThreshold=100"]),
                         ("FindAlignmentCandidatesLowHash0.py", []), ("ComputeAlignments.py", ["minAlignedMarkerCount=100"]),
                         ("CreateReadGraph.py", ["maxAlignmentCount=6"])):
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", script)] + args, cwd=str(tmp_path), env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, script + ": " + out.stdout[-1500:] + out.stderr[-1500:]
        if script == "FlagPalindromicReads.py":
            assert "Flagged 0 reads as palindromic out of 20 total." in out.stdout
    stored, _ = shim.open_vector(os.path.join(d, "AlignmentCandidates"), 12)
    assert np.array_equal(stored.view("<u4").reshape(-1, 3)[:, :2], g.z["lh0_candidates"][:, :2])
    al = oracle_lib.align4_batch(g.toc, g.data7, g.candidates(0), abi.default_align4_options(), want_ordinals=False, threads=0)
    rows, _ = shim.open_vector(os.path.join(d, "AlignmentData"), 64)
    assert len(rows) == len(al.alignment_data) > 50
    assert os.path.exists(os.path.join(d, "ReadGraphEdges")) and os.path.exists(os.path.join(d, "CandidateTable.toc"))
