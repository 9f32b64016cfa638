"""TEST INFRASTRUCTURE ONLY.

ctypes bindings for the two checker libraries:

  RefLib     oracle/_ref/libshasta_ref.so   the reference's own LowHash0 / Align4 control
                                            flow compiled in place (oracle/ref_build/)
  OracleLib  oracle/_build/liboracle.so     the CPU restatement in this directory

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing under shasta_amd/ does.
"""
import ctypes as C
import os
import tempfile

import numpy as np

from shasta_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "libshasta_ref.so")
ORACLE_SO = os.path.join(HERE, "_build", "liboracle.so")


def ref_available():
    return os.path.exists(REF_SO)


def oracle_available():
    return os.path.exists(ORACLE_SO)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class _Base:
    prefix = ""

    def __init__(self, path):
        self.lib = C.CDLL(path)
        self.path = path
        f = getattr(self.lib, self.prefix + "last_error")
        f.restype = C.c_char_p
        self._last_error = f

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self._last_error().decode()))

    def set_tie_policy(self, index):
        """The DP tie policy of this library's shimmed / restated SeqAn call (oracle/banded_dp.hpp, tiePolicyByIndex):
        0 = the restated reading (diagonal >= vertical >= horizontal, first maximum); 1..11 the alternatives."""
        getattr(self.lib, self.prefix + "set_tie_policy")(C.c_int(index))

    def _align4(self, toc, data7, candidates, options, want_ordinals):
        toc = _u64(toc)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        candidates = np.ascontiguousarray(candidates, dtype=abi.PAIR_DTYPE)
        read_count = (len(toc) - 1) // 2
        res = abi.Align4Result()
        fn = getattr(self.lib, self.prefix + "align4_batch")
        fn.restype = C.c_int
        rc = fn(C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
                C.c_uint64(len(candidates)), C.c_void_p(candidates.ctypes.data),
                C.byref(options), C.c_int(1 if want_ordinals else 0), C.byref(res))
        self._check(rc, self.prefix + "align4_batch")
        out = abi.Align4Output(res, len(candidates), want_ordinals)
        getattr(self.lib, self.prefix + "align4_free")(C.byref(res))
        return out


class RefLib(_Base):
    prefix = "ref_"

    def __init__(self):
        super().__init__(REF_SO)
        self.lib.ref_murmur64a.restype = C.c_uint64

    def murmur64a(self, data, seed):
        b = bytes(data)
        return int(self.lib.ref_murmur64a(b, C.c_int(len(b)), C.c_uint64(seed)))

    def test_alignment_compression(self):
        self._check(self.lib.ref_test_alignment_compression(), "testAlignmentCompression")

    def compress(self, ordinals):
        o = np.ascontiguousarray(ordinals, dtype=np.uint32).reshape(-1, 2)
        p = C.POINTER(C.c_uint8)()
        n = C.c_uint64()
        self._check(self.lib.ref_compress(abi.as_ptr(o, C.c_uint32), C.c_uint64(len(o)),
                                          C.byref(p), C.byref(n)), "ref_compress")
        out = abi.copy_array(p, n.value, "u1")
        self.lib.ref_free(p)
        return out

    def decompress(self, data):
        d = np.ascontiguousarray(data, dtype=np.uint8)
        p = C.POINTER(C.c_uint32)()
        n = C.c_uint64()
        self._check(self.lib.ref_decompress(abi.as_ptr(d, C.c_uint8), C.c_uint64(len(d)),
                                            C.byref(p), C.byref(n)), "ref_decompress")
        out = abi.copy_array(p, 2 * n.value, "<u4").reshape(-1, 2)
        self.lib.ref_free(p)
        return out

    def flag_palindromic_reads(self, toc, data7, max_skip=100, max_drift=100, max_marker_frequency=10,
                               aligned_fraction_threshold=0.1, near_diagonal_fraction_threshold=0.1, delta_threshold=100, threads=1):
        """The reference's method-0 self-alignment of every read + the flag rule (src/AssemblerAlign.cpp:702-770).
        Returns (flags u8[R], aligned marker count u32[R], near-diagonal marker count u32[R], FNV-1a digest of the
        aligned ordinals u64[R])."""
        toc = _u64(toc)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        read_count = (len(toc) - 1) // 2
        flags = np.zeros(read_count, np.uint8)
        aligned = np.zeros(read_count, np.uint32)
        near = np.zeros(read_count, np.uint32)
        digests = np.zeros(read_count, np.uint64)
        rc = self.lib.ref_flag_palindromic_reads(
            C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
            C.c_uint32(max_skip), C.c_uint32(max_drift), C.c_uint32(max_marker_frequency),
            C.c_double(aligned_fraction_threshold), C.c_double(near_diagonal_fraction_threshold), C.c_uint32(delta_threshold),
            C.c_uint64(threads), abi.as_ptr(flags, C.c_uint8), abi.as_ptr(aligned, C.c_uint32), abi.as_ptr(near, C.c_uint32),
            abi.as_ptr(digests, C.c_uint64))
        if rc:
            self.lib.ref_palindromic_last_error.restype = C.c_char_p
            raise RuntimeError("ref_flag_palindromic_reads failed: %s" % self.lib.ref_palindromic_last_error().decode())
        return flags, aligned, near, digests

    def suppress_alignment_flags(self, fasta_path, data_directory, read_id0, read_id1, delta):
        """Loads the FASTA with the reference's ReadLoader into Reads files under data_directory and returns
        (Assembler::suppressAlignment of every pair as u8, read count)."""
        r0 = np.ascontiguousarray(read_id0, np.uint32)
        r1 = np.ascontiguousarray(read_id1, np.uint32)
        out = np.zeros(len(r0), np.uint8)
        count = C.c_uint64()
        rc = self.lib.ref_suppress_alignment_flags(fasta_path.encode(), data_directory.encode(), C.c_uint64(len(r0)),
                                                   abi.as_ptr(r0, C.c_uint32), abi.as_ptr(r1, C.c_uint32), C.c_uint64(delta),
                                                   abi.as_ptr(out, C.c_uint8), C.byref(count))
        if rc:
            self.lib.ref_suppress_last_error.restype = C.c_char_p
            raise RuntimeError("ref_suppress_alignment_flags failed: %s" % self.lib.ref_suppress_last_error().decode())
        return out, int(count.value)

    def alignment_info(self, ordinals, nx, ny):
        o = np.ascontiguousarray(ordinals, dtype=np.uint32).reshape(-1, 2)
        info = abi.AlignmentInfo()
        self._check(self.lib.ref_alignment_info(abi.as_ptr(o, C.c_uint32), C.c_uint64(len(o)),
                                                C.c_uint32(nx), C.c_uint32(ny), C.byref(info)),
                    "ref_alignment_info")
        return info

    def markers_from_fasta(self, path, k=10, probability=0.1, seed=231, min_read_length=10000, threads=0):
        rc_ = C.c_uint64()
        toc = C.POINTER(C.c_uint64)()
        data = C.POINTER(C.c_uint8)()
        self._check(self.lib.ref_markers_from_fasta(
            path.encode(), C.c_uint64(k), C.c_double(probability), C.c_int(seed),
            C.c_uint64(min_read_length), C.c_uint64(threads),
            C.byref(rc_), C.byref(toc), C.byref(data)), "ref_markers_from_fasta")
        n = 2 * rc_.value
        toc_a = abi.copy_array(toc, n + 1, "<u8")
        data_a = abi.copy_array(data, 7 * int(toc_a[-1]), "u1")
        self.lib.ref_free(toc)
        self.lib.ref_free(data)
        return toc_a, data_a

    def reads_and_markers_from_fasta(self, path, k=10, probability=0.1, seed=231, min_read_length=10000, threads=0):
        """-> dict(reads_toc, reads_data, base_counts, is_marker, toc, data7): what the reference's
        MarkerFinder read (reads as stored, isMarker flags) and what it wrote."""
        rc_ = C.c_uint64()
        rt, rd, bc = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        im, toc, data = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint8)()
        self._check(self.lib.ref_reads_and_markers_from_fasta(
            path.encode(), C.c_uint64(k), C.c_double(probability), C.c_int(seed), C.c_uint64(min_read_length), C.c_uint64(threads),
            C.byref(rc_), C.byref(rt), C.byref(rd), C.byref(bc), C.byref(im), C.byref(toc), C.byref(data)),
            "ref_reads_and_markers_from_fasta")
        r = rc_.value
        out = {"reads_toc": abi.copy_array(rt, r + 1, "<u8")}
        out["reads_data"] = abi.copy_array(rd, int(out["reads_toc"][-1]), "<u8")
        out["base_counts"] = abi.copy_array(bc, r, "<u8")
        out["is_marker"] = abi.copy_array(im, 1 << (2 * k), "u1")
        out["toc"] = abi.copy_array(toc, 2 * r + 1, "<u8")
        out["data7"] = abi.copy_array(data, 7 * int(out["toc"][-1]), "u1")
        for ptr in (rt, rd, bc, im, toc, data):
            self.lib.ref_free(ptr)
        return out

    # --- Data/ directory fixtures through the reference's own containers ---
    def write_data_dir(self, directory, toc, data7, flags=None):
        toc = _u64(toc)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        read_count = (len(toc) - 1) // 2
        fp = abi.as_ptr(np.ascontiguousarray(flags, np.uint8), C.c_uint8) if flags is not None else None
        self._check(self.lib.ref_write_data_dir(directory.encode(), C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64),
                                                C.c_void_p(data7.ctypes.data), fp), "ref_write_data_dir")

    def open_vector(self, path, object_size):
        """-> (data as uint8[objectCount, objectSize], file size implied by the header)."""
        count, size = C.c_uint64(), C.c_uint64()
        self._check(self.lib.ref_open_vector(path.encode(), C.c_uint64(object_size), C.byref(count), C.byref(size), None, C.c_uint64(0)), "ref_open_vector")
        out = np.zeros((count.value, object_size), dtype=np.uint8)
        self._check(self.lib.ref_open_vector(path.encode(), C.c_uint64(object_size), C.byref(count), C.byref(size),
                                             C.c_void_p(out.ctypes.data), C.c_uint64(out.nbytes)), "ref_open_vector")
        return out, int(size.value)

    def lowhash0_files(self, directory, params, work_dir, threads=1):
        self._check(self.lib.ref_lowhash0_files(directory.encode(), C.byref(params), C.c_uint64(threads), work_dir.encode()), "ref_lowhash0_files")

    def store_alignments(self, directory, alignment_data, compressed_toc, compressed_data):
        rows = np.ascontiguousarray(alignment_data)
        toc = _u64(compressed_toc)
        data = np.ascontiguousarray(compressed_data, dtype=np.uint8)
        self._check(self.lib.ref_store_alignments(directory.encode(), C.c_uint64(len(rows)), C.c_void_p(rows.ctypes.data),
                                                  abi.as_ptr(toc, C.c_uint64), C.c_void_p(data.ctypes.data)), "ref_store_alignments")

    def lowhash0(self, toc, data7, flags, params, threads=0, work_dir=None):
        toc = _u64(toc)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        read_count = (len(toc) - 1) // 2
        flags = np.zeros(read_count, np.uint8) if flags is None else np.ascontiguousarray(flags, np.uint8)
        stats = np.zeros((read_count, 3), dtype=np.uint64)
        res = abi.LowHash0Result()
        with tempfile.TemporaryDirectory() as tmp:
            wd = work_dir or tmp
            rc = self.lib.ref_lowhash0(
                C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
                abi.as_ptr(flags, C.c_uint8), C.byref(params), C.c_uint64(threads), wd.encode(),
                abi.as_ptr(stats, C.c_uint64), C.byref(res))
            self._check(rc, "ref_lowhash0")
            out = abi.LowHash0Output(res, stats)
            if work_dir is None:
                with open(os.path.join(tmp, "LowHashBucketHistogram.csv"), "rb") as f:
                    out.histogram_csv = f.read()
        self.lib.ref_lowhash0_free(C.byref(res))
        return out

    def align4_batch(self, toc, data7, candidates, options, want_ordinals=True, threads=1):
        """threads=1 reproduces the reference's single-thread output order; any thread count
        gives the same rows here because results are gathered in candidate order."""
        toc = _u64(toc)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        candidates = np.ascontiguousarray(candidates, dtype=abi.PAIR_DTYPE)
        read_count = (len(toc) - 1) // 2
        res = abi.Align4Result()
        rc = self.lib.ref_align4_batch_mt(
            C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
            C.c_uint64(len(candidates)), C.c_void_p(candidates.ctypes.data),
            C.byref(options), C.c_int(1 if want_ordinals else 0), C.c_uint64(threads), C.byref(res))
        self._check(rc, "ref_align4_batch_mt")
        out = abi.Align4Output(res, len(candidates), want_ordinals)
        self.lib.ref_align4_free(C.byref(res))
        return out


    def compute_sorted_markers(self, toc, data7, threads=0):
        """Assembler::computeSortedMarkers (src/AssemblerAlign4.cpp:190-261) in the reference's container, once for all oriented
        reads, as computeAlignments does before its threads start; align4_batch calls on the SAME data7 array then read it
        instead of sorting the reads of every candidate.  -> seconds."""
        toc = _u64(toc)
        assert data7.dtype == np.uint8 and data7.flags["C_CONTIGUOUS"]            # (kept by address: the caller's array itself)
        seconds = C.c_double(0.0)
        self._check(self.lib.ref_compute_sorted_markers(C.c_uint64((len(toc) - 1) // 2), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
                                                        C.c_uint64(threads), C.byref(seconds)), "ref_compute_sorted_markers")
        return seconds.value

    def drop_sorted_markers(self):
        self.lib.ref_drop_sorted_markers()

    def alignment_table(self, alignment_data, read_count):
        """Assembler::computeAlignmentTable (src/AssemblerAlign.cpp:509-571) -> (toc uint64[2 R + 1], values uint32[4 N], seconds)."""
        rows = np.ascontiguousarray(alignment_data)
        assert rows.dtype.itemsize == 64
        toc = np.zeros(2 * int(read_count) + 1, np.uint64)
        values = np.zeros(max(1, 4 * len(rows)), np.uint32)
        seconds = C.c_double(0.0)
        self._check(self.lib.ref_alignment_table(C.c_uint64(read_count), C.c_uint64(len(rows)), C.c_void_p(rows.ctypes.data if len(rows) else None),
                                                 abi.as_ptr(toc, C.c_uint64), abi.as_ptr(values, C.c_uint32), C.byref(seconds)), "ref_alignment_table")
        return toc, values[:4 * len(rows)], seconds.value

    def align3_batch(self, toc, data7, candidates, options, want_ordinals=True, threads=1):
        """Align method 3 through the reference's own Assembler::alignOrientedReads3."""
        toc = _u64(toc)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        candidates = np.ascontiguousarray(candidates, dtype=abi.PAIR_DTYPE)
        read_count = (len(toc) - 1) // 2
        res = abi.Align4Result()
        rc = self.lib.ref_align3_batch_mt(
            C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
            C.c_uint64(len(candidates)), C.c_void_p(candidates.ctypes.data),
            C.byref(options), C.c_int(1 if want_ordinals else 0), C.c_uint64(threads), C.byref(res))
        self._check(rc, "ref_align3_batch_mt")
        out = abi.Align4Output(res, len(candidates), want_ordinals)
        self.lib.ref_align4_free(C.byref(res))
        return out

    def kmer_hashes(self, k):
        """KmerInfo::hash of all 4^k k-mer ids (src/AssemblerKmers.cpp:182-186)."""
        out = np.zeros(1 << (2 * k), dtype=np.uint32)
        self._check(self.lib.ref_kmer_hashes(C.c_uint64(k), abi.as_ptr(out, C.c_uint32)), "ref_kmer_hashes")
        return out


class OracleLib(_Base):
    prefix = "oracle_"

    def __init__(self):
        super().__init__(ORACLE_SO)
        self.lib.oracle_murmur64a.restype = C.c_uint64

    def murmur64a(self, data, seed):
        b = bytes(data)
        return int(self.lib.oracle_murmur64a(b, C.c_int(len(b)), C.c_uint64(seed)))

    def flag_palindromic_reads(self, toc, data7, max_skip=100, max_drift=100, max_marker_frequency=10,
                               aligned_fraction_threshold=0.1, near_diagonal_fraction_threshold=0.1, delta_threshold=100):
        """Restated method 0 + flag rule (oracle/method0.hpp); same returns as RefLib.flag_palindromic_reads."""
        toc = _u64(toc)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        read_count = (len(toc) - 1) // 2
        flags = np.zeros(read_count, np.uint8)
        aligned = np.zeros(read_count, np.uint32)
        near = np.zeros(read_count, np.uint32)
        digests = np.zeros(read_count, np.uint64)
        self._check(self.lib.oracle_flag_palindromic_reads(
            C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
            C.c_uint32(max_skip), C.c_uint32(max_drift), C.c_uint32(max_marker_frequency),
            C.c_double(aligned_fraction_threshold), C.c_double(near_diagonal_fraction_threshold), C.c_uint32(delta_threshold),
            abi.as_ptr(flags, C.c_uint8), abi.as_ptr(aligned, C.c_uint32), abi.as_ptr(near, C.c_uint32),
            abi.as_ptr(digests, C.c_uint64)), "oracle_flag_palindromic_reads")
        return flags, aligned, near, digests

    def hash_windows(self, kmer_ids, m, iteration):
        k = np.ascontiguousarray(kmer_ids, dtype=np.uint32)
        n = len(k)
        out = np.zeros(max(0, n - m + 1), dtype=np.uint64)
        self._check(self.lib.oracle_hash_windows(abi.as_ptr(k, C.c_uint32), C.c_uint64(n), C.c_uint64(m),
                                                 C.c_uint64(iteration), abi.as_ptr(out, C.c_uint64)),
                    "oracle_hash_windows")
        return out

    def lowhash0(self, toc, data7, flags, params, threads=1):
        toc = _u64(toc)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        read_count = (len(toc) - 1) // 2
        flags = np.zeros(read_count, np.uint8) if flags is None else np.ascontiguousarray(flags, np.uint8)
        stats = np.zeros((read_count, 3), dtype=np.uint64)
        res = abi.LowHash0Result()
        rc = self.lib.oracle_lowhash0(
            C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
            abi.as_ptr(flags, C.c_uint8), C.byref(params), C.c_uint64(threads),
            abi.as_ptr(stats, C.c_uint64), C.byref(res))
        self._check(rc, "oracle_lowhash0")
        out = abi.LowHash0Output(res, stats)
        self.lib.oracle_lowhash0_free(C.byref(res))
        return out

    def align4_batch(self, toc, data7, candidates, options, want_ordinals=True, threads=1):
        self.lib.oracle_set_threads(C.c_uint64(threads))
        return self._align4(toc, data7, candidates, options, want_ordinals)

    def align3_batch(self, toc, data7, candidates, options, want_ordinals=True, threads=1):
        self.lib.oracle_set_threads(C.c_uint64(threads))
        toc = _u64(toc)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        candidates = np.ascontiguousarray(candidates, dtype=abi.PAIR_DTYPE)
        read_count = (len(toc) - 1) // 2
        res = abi.Align4Result()
        rc = self.lib.oracle_align3_batch(
            C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
            C.c_uint64(len(candidates)), C.c_void_p(candidates.ctypes.data),
            C.byref(options), C.c_int(1 if want_ordinals else 0), C.byref(res))
        self._check(rc, "oracle_align3_batch")
        out = abi.Align4Output(res, len(candidates), want_ordinals)
        self.lib.oracle_align4_free(C.byref(res))
        return out

    def align3_stages(self, k0, k1, options):
        """-> dict of the stage products of method 3 for one pair of kmer-id sequences."""
        k0 = np.ascontiguousarray(k0, dtype=np.uint32)
        k1 = np.ascontiguousarray(k1, dtype=np.uint32)
        out = np.zeros(8, dtype=np.int64)
        self._check(self.lib.oracle_align3_stages(
            abi.as_ptr(k0, C.c_uint32), C.c_uint32(len(k0)), abi.as_ptr(k1, C.c_uint32), C.c_uint32(len(k1)),
            C.byref(options), abi.as_ptr(out, C.c_int64)), "oracle_align3_stages")
        names = ["downsampled0", "downsampled1", "aligned", "offsetMin", "offsetMax", "bandMin", "bandMax", "bandTooWide"]
        return dict(zip(names, (int(v) for v in out)))

    def kmer_hashes(self, kmer_ids, k):
        ids = np.ascontiguousarray(kmer_ids, dtype=np.uint32)
        out = np.zeros(len(ids), dtype=np.uint32)
        self._check(self.lib.oracle_kmer_hashes(abi.as_ptr(ids, C.c_uint32), C.c_uint64(len(ids)), C.c_uint64(k),
                                                abi.as_ptr(out, C.c_uint32)), "oracle_kmer_hashes")
        return out

    def find_markers(self, reads_toc, reads_data, base_counts, k, is_marker):
        """-> (toc uint64[2R+1], data7 uint8[7*M])."""
        rt, rd, bc = _u64(reads_toc), _u64(reads_data), _u64(base_counts)
        im = np.ascontiguousarray(is_marker, dtype=np.uint8)
        toc = np.zeros(2 * len(bc) + 1, dtype=np.uint64)
        p = C.POINTER(C.c_uint8)()
        self._check(self.lib.oracle_find_markers(C.c_uint64(len(bc)), abi.as_ptr(rt, C.c_uint64), abi.as_ptr(rd, C.c_uint64),
                                                 abi.as_ptr(bc, C.c_uint64), C.c_uint64(k), abi.as_ptr(im, C.c_uint8),
                                                 abi.as_ptr(toc, C.c_uint64), C.byref(p)), "oracle_find_markers")
        data = abi.copy_array(p, 7 * int(toc[-1]), "u1")
        self.lib.oracle_free(p)
        return toc, data

    def banded_dp(self, k0, k1, band_min, band_max):
        k0 = np.ascontiguousarray(k0, dtype=np.uint32)
        k1 = np.ascontiguousarray(k1, dtype=np.uint32)
        cap = min(len(k0), len(k1)) + 1
        out = np.zeros((cap, 2), dtype=np.uint32)
        count = C.c_uint64()
        score = C.c_int32()
        self._check(self.lib.oracle_banded_dp(
            abi.as_ptr(k0, C.c_uint32), C.c_uint32(len(k0)), abi.as_ptr(k1, C.c_uint32), C.c_uint32(len(k1)),
            C.c_int32(band_min), C.c_int32(band_max), abi.as_ptr(out, C.c_uint32), C.c_uint64(cap),
            C.byref(count), C.byref(score)), "oracle_banded_dp")
        return out[:count.value].copy(), score.value

    def sparse_dp(self, k0, k1, band_min, band_max, scan_budget=64):
        """The same task from its matches (oracle/sparse_chain.hpp) -> (ordinals [n, 2], dict(certified, score, hits, scan_steps, reason))."""
        k0 = np.ascontiguousarray(k0, dtype=np.uint32)
        k1 = np.ascontiguousarray(k1, dtype=np.uint32)
        cap = min(len(k0), len(k1)) + 1
        out = np.zeros(2 * cap, dtype=np.uint32)
        count = C.c_uint64()
        info = np.zeros(5, dtype=np.int64)
        self._check(self.lib.oracle_sparse_dp(abi.as_ptr(k0, C.c_uint32), C.c_uint32(len(k0)), abi.as_ptr(k1, C.c_uint32), C.c_uint32(len(k1)),
                                              C.c_int32(band_min), C.c_int32(band_max), C.c_uint32(scan_budget), abi.as_ptr(out, C.c_uint32), C.c_uint64(cap),
                                              C.byref(count), abi.as_ptr(info, C.c_int64)), "oracle_sparse_dp")
        return out[:2 * count.value].reshape(-1, 2).copy(), dict(zip(("certified", "score", "hits", "scan_steps", "reason"), (int(v) for v in info)))

    def anchored_dp(self, k0, k1, band_min, band_max):
        """The task with the dense DP confined to where the optimal chains differ (oracle/anchored_chain.hpp), under the policy in
        force -> (ordinals [n, 2], dict(score, hits, anchors, windows, dense_cells, whole_task_dense))."""
        k0 = np.ascontiguousarray(k0, dtype=np.uint32)
        k1 = np.ascontiguousarray(k1, dtype=np.uint32)
        cap = min(len(k0), len(k1)) + 1
        out = np.zeros(2 * cap, dtype=np.uint32)
        count = C.c_uint64()
        info = np.zeros(7, dtype=np.int64)
        self._check(self.lib.oracle_anchored_dp(abi.as_ptr(k0, C.c_uint32), C.c_uint32(len(k0)), abi.as_ptr(k1, C.c_uint32), C.c_uint32(len(k1)),
                                                C.c_int32(band_min), C.c_int32(band_max), abi.as_ptr(out, C.c_uint32), C.c_uint64(cap),
                                                C.byref(count), abi.as_ptr(info, C.c_int64)), "oracle_anchored_dp")
        return out[:2 * count.value].reshape(-1, 2).copy(), dict(zip(("score", "hits", "anchors", "windows", "dense_cells", "whole_task_dense", "largest_window"), (int(v) for v in info)))

    def sparse_census(self, on=None, reset=False, scan_budget=None, one_hit_per_marker=False, running_max_bound=False, anchored=None):
        """Bookkeeping of the sparse path's prototype over the DP tasks align4_batch runs (oracle.cpp: oracle_sparse_census_read)."""
        if scan_budget is not None:
            self.lib.oracle_sparse_census_options(C.c_uint32(scan_budget), C.c_int(1 if one_hit_per_marker else 0), C.c_int(1 if running_max_bound else 0))
        if anchored is not None:
            self.lib.oracle_sparse_census_anchored(C.c_int(1 if anchored else 0))
        if reset:
            self.lib.oracle_sparse_census_reset()
        if on is not None:
            self.lib.oracle_sparse_census(C.c_int(1 if on else 0))
        out = np.zeros(22, dtype=np.uint64)
        self.lib.oracle_sparse_census_read(abi.as_ptr(out, C.c_uint64))
        names = ("tasks", "certified", "certified_but_different", "dense_cells", "hits", "scan_steps", "reason_certified", "reason_several_chains",
                 "reason_ties_with_empty", "reason_scan_budget", "dense_cells_of_certified", "aligned_pairs", "reason_two_hits_of_one_marker",
                 "anchored_different", "anchored_dense_cells", "anchored_windows", "anchored_whole_tasks", "anchors", "anchored_largest_window", "anchored_tasks_with_a_window_over_16384",
                 "ambiguous_tasks_with_a_live_link_over_29_hits", "anchored_tasks_with_over_128_windows")
        return dict(zip(names, (int(v) for v in out)))

    def compress(self, ordinals):
        o = np.ascontiguousarray(ordinals, dtype=np.uint32).reshape(-1, 2)
        p = C.POINTER(C.c_uint8)()
        n = C.c_uint64()
        self._check(self.lib.oracle_compress(abi.as_ptr(o, C.c_uint32), C.c_uint64(len(o)),
                                             C.byref(p), C.byref(n)), "oracle_compress")
        out = abi.copy_array(p, n.value, "u1")
        self.lib.oracle_free(p)
        return out

    def decompress(self, data):
        d = np.ascontiguousarray(data, dtype=np.uint8)
        p = C.POINTER(C.c_uint32)()
        n = C.c_uint64()
        self._check(self.lib.oracle_decompress(abi.as_ptr(d, C.c_uint8), C.c_uint64(len(d)),
                                               C.byref(p), C.byref(n)), "oracle_decompress")
        out = abi.copy_array(p, 2 * n.value, "<u4").reshape(-1, 2)
        self.lib.oracle_free(p)
        return out

    def alignment_info(self, ordinals, nx, ny):
        o = np.ascontiguousarray(ordinals, dtype=np.uint32).reshape(-1, 2)
        info = abi.AlignmentInfo()
        self._check(self.lib.oracle_alignment_info(abi.as_ptr(o, C.c_uint32), C.c_uint64(len(o)),
                                                   C.c_uint32(nx), C.c_uint32(ny), C.byref(info)),
                    "oracle_alignment_info")
        return info
