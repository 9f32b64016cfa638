"""ctypes access to the host layer's file classes (tests/host_shim/test_shim.cpp) for the tests."""
import ctypes as C
import os

import numpy as np

from shasta_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "shasta_amd", "_build", "libshasta_host_shim.so")
EMU_SHIM = os.path.join(ROOT, "tests", "emu", "_build", "libshasta_host_shim_emu.so")       # the same on the wave64 emulator (no GPU needed)
EMU_HOST_SO = os.path.join(ROOT, "tests", "emu", "_build", "libshasta_mi355x_host_emu.so")
STAGE = os.path.join(ROOT, "shasta_amd", "_build", "shasta_mi355x_stage")


def emulated_build():
    """Builds tests/emu (kernels on CPU fibers + the host layer linked against them) if it is missing."""
    if not (os.path.exists(EMU_SHIM) and os.path.exists(EMU_HOST_SO)):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu"), "-j8"])


class HostShim:
    def __init__(self, path=SHIM):
        self.lib = C.CDLL(path)
        self.lib.host_last_error.restype = C.c_char_p

    def _check(self, rc, what):
        if rc:
            raise RuntimeError("%s failed: %s" % (what, self.lib.host_last_error().decode()))

    def write_data_dir(self, directory, toc, data7, flags=None):
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        fp = abi.as_ptr(np.ascontiguousarray(flags, np.uint8), C.c_uint8) if flags is not None else None
        self._check(self.lib.host_write_data_dir(directory.encode(), C.c_uint64((len(toc) - 1) // 2), abi.as_ptr(toc, C.c_uint64),
                                                 C.c_void_p(data7.ctypes.data), fp), "host_write_data_dir")

    def write_read_flags(self, directory, read_count, flags=None):
        fp = abi.as_ptr(np.ascontiguousarray(flags, np.uint8), C.c_uint8) if flags is not None else None
        self._check(self.lib.host_write_read_flags(directory.encode(), C.c_uint64(read_count), fp), "host_write_read_flags")

    def open_vector(self, path, object_size):
        count, size = C.c_uint64(), C.c_uint64()
        self._check(self.lib.host_open_vector(path.encode(), C.c_uint64(object_size), C.byref(count), C.byref(size), None, C.c_uint64(0)), "host_open_vector")
        out = np.zeros((count.value, object_size), dtype=np.uint8)
        self._check(self.lib.host_open_vector(path.encode(), C.c_uint64(object_size), C.byref(count), C.byref(size),
                                              C.c_void_p(out.ctypes.data), C.c_uint64(out.nbytes)), "host_open_vector")
        return out, int(size.value)

    def store_alignments(self, directory, alignment_data, compressed_toc, compressed_data):
        rows = np.ascontiguousarray(alignment_data)
        toc = np.ascontiguousarray(compressed_toc, dtype=np.uint64)
        data = np.ascontiguousarray(compressed_data, dtype=np.uint8)
        self._check(self.lib.host_store_alignments(directory.encode(), C.c_uint64(len(rows)), C.c_void_p(rows.ctypes.data),
                                                   abi.as_ptr(toc, C.c_uint64), C.c_void_p(data.ctypes.data)), "host_store_alignments")

    def write_kmers(self, directory, k, is_marker=None):
        im = abi.as_ptr(np.ascontiguousarray(is_marker, np.uint8), C.c_uint8) if is_marker is not None else None
        self._check(self.lib.host_write_kmers(directory.encode(), C.c_uint64(k), im), "host_write_kmers")

    def write_reads(self, directory, reads_toc, reads_data, base_counts):
        rt = np.ascontiguousarray(reads_toc, np.uint64)
        rd = np.ascontiguousarray(reads_data, np.uint64)
        bc = np.ascontiguousarray(base_counts, np.uint64)
        self._check(self.lib.host_write_reads(directory.encode(), C.c_uint64(len(bc)), abi.as_ptr(rt, C.c_uint64),
                                              abi.as_ptr(rd, C.c_uint64), abi.as_ptr(bc, C.c_uint64)), "host_write_reads")

    def store_candidates(self, directory, candidates):
        c = np.ascontiguousarray(candidates, dtype=abi.PAIR_DTYPE)
        self._check(self.lib.host_store_candidates(directory.encode(), C.c_uint64(len(c)), C.c_void_p(c.ctypes.data)), "host_store_candidates")

    def compute_candidate_table(self, directory, read_count):
        self._check(self.lib.host_compute_candidate_table(directory.encode(), C.c_uint64(read_count)), "host_compute_candidate_table")

    def create_read_graph(self, directory, max_alignment_count):
        kept = C.c_uint64()
        self._check(self.lib.host_create_read_graph(directory.encode(), C.c_uint32(max_alignment_count), C.byref(kept)), "host_create_read_graph")
        return int(kept.value)

    def compute_alignment_table(self, directory, read_count):
        self._check(self.lib.host_compute_alignment_table(directory.encode(), C.c_uint64(read_count)), "host_compute_alignment_table")


def alignment_table_expected(read_count, alignment_data, index_dtype=np.uint32):
    """Assembler::computeAlignmentTable (src/AssemblerAlign.cpp:509-571) and
    AlignmentCandidates::computeCandidateTable (src/AssemblerAlignmentCandidates.cpp:388-447) in python:
    per oriented read, the indices of the pairs it takes part in (directly or reverse complemented),
    sorted by (other oriented read, index)."""
    sections = [[] for _ in range(2 * read_count)]
    for i, ad in enumerate(alignment_data):
        o0 = int(ad["readId0"]) << 1
        o1 = (int(ad["readId1"]) << 1) | (0 if ad["isSameStrand"] else 1)
        for a, b in ((o0, o1), (o1, o0), (o0 ^ 1, o1 ^ 1), (o1 ^ 1, o0 ^ 1)):
            sections[a].append((b, i))
    toc = np.zeros(2 * read_count + 1, dtype=index_dtype)
    data = []
    for k, sec in enumerate(sections):
        sec.sort()
        data += [i for _, i in sec]
        toc[k + 1] = len(data)
    return toc, np.asarray(data, dtype=index_dtype)


def read_graph_expected(read_count, alignment_data, max_alignment_count):
    """Assembler::createReadGraph + createReadGraphUsingSelectedAlignments (src/AssemblerReadGraph.cpp:35-157)
    in python: kept flags, edges (oriented read pair, alignment id) and connectivity rows."""
    per_read = [[] for _ in range(read_count)]
    for i, ad in enumerate(alignment_data):
        per_read[int(ad["readId0"])].append((int(ad["markerCount"]), i))
        per_read[int(ad["readId1"])].append((int(ad["markerCount"]), i))
    keep = np.zeros(len(alignment_data), bool)
    for lst in per_read:
        for _, i in sorted(lst, reverse=True)[:max_alignment_count]:
            keep[i] = True
    edges = []
    for i, ad in enumerate(alignment_data):
        if not keep[i]:
            continue
        o0 = int(ad["readId0"]) << 1
        o1 = (int(ad["readId1"]) << 1) | (0 if ad["isSameStrand"] else 1)
        edges.append((o0, o1, i))
        edges.append((o0 ^ 1, o1 ^ 1, i))
    rows = [[] for _ in range(2 * read_count)]
    for e, (a, b, _) in enumerate(edges):
        rows[a].append(e)
        rows[b].append(e)
    toc = np.zeros(2 * read_count + 1, np.uint32)
    data = []
    for k, row in enumerate(rows):
        data += sorted(row, reverse=True)                 # VectorOfVectors::store fills a row from the back
        toc[k + 1] = len(data)
    return keep, edges, toc, np.asarray(data, np.uint32)
