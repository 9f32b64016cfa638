"""ctypes loader for shasta_amd/_build/libshasta_mi355x.so.

There is no CPU fallback: if the HIP library is missing, or no gfx950 device is
usable, every compute call raises.  (The CPU oracle under oracle/ is test
infrastructure and is never imported from here.)
"""
import ctypes as C
import os

import numpy as np

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "_build", "libshasta_mi355x.so")

# Every symbol include/shasta_mi355x.h declares.
EXPORTS = [
    "shasta_mi355x_last_error", "shasta_mi355x_version", "shasta_mi355x_device_count", "shasta_mi355x_environment_warnings",
    "shasta_mi355x_lowhash0", "shasta_mi355x_lowhash0_free",
    "shasta_mi355x_align4_batch", "shasta_mi355x_align4_free",
    "shasta_mi355x_create", "shasta_mi355x_destroy",
    "shasta_mi355x_set_markers", "shasta_mi355x_set_kmer_ids",
    "shasta_mi355x_lowhash0_run", "shasta_mi355x_align4_run", "shasta_mi355x_align4_run_borrowed", "shasta_mi355x_kernel_table", "shasta_mi355x_kernel_table_reset",
    "shasta_mi355x_hash_windows", "shasta_mi355x_banded_dp", "shasta_mi355x_banded_dp_many", "shasta_mi355x_calibrate",
    "shasta_mi355x_pair_table", "shasta_mi355x_read_graph_keep", "shasta_mi355x_alignment_table",
    "shasta_mi355x_set_kmer_ids_device", "shasta_mi355x_memcpy", "shasta_mi355x_free",
    "shasta_mi355x_lh_begin", "shasta_mi355x_lh_one_pass_fits", "shasta_mi355x_lh_hash", "shasta_mi355x_lh_buckets", "shasta_mi355x_lh_merge",
    "shasta_mi355x_lh_finish", "shasta_mi355x_lh_finish_on_device", "shasta_mi355x_lh_hash_all", "shasta_mi355x_lh_buckets_all", "shasta_mi355x_lh_merge_all",
    "shasta_mi355x_align3_run", "shasta_mi355x_align3_batch",
    "shasta_mi355x_find_markers", "shasta_mi355x_find_markers_free",
    "shasta_mi355x_palindromic_screen",
    "shasta_mi355x_group_create", "shasta_mi355x_group_destroy", "shasta_mi355x_group_set_markers", "shasta_mi355x_group_set_kmer_ids",
    "shasta_mi355x_group_lowhash0_run", "shasta_mi355x_group_align4_run", "shasta_mi355x_group_align3_run",
    "shasta_mi355x_group_align4_run_borrowed", "shasta_mi355x_group_align3_run_borrowed",
    "shasta_mi355x_lowhash0_multi", "shasta_mi355x_align4_batch_multi", "shasta_mi355x_align3_batch_multi",
]


class LibraryNotBuilt(RuntimeError):
    pass


class Library:
    def __init__(self, path=SO_PATH):
        if not os.path.exists(path):
            raise LibraryNotBuilt(
                "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C shasta_amd/csrc). There is no CPU fallback." % path)
        self.path = path
        # The aligner keeps six workers' streams busy; the HIP runtime deals streams to GPU_MAX_HW_QUEUES hardware queues (default 4)
        # and reads the variable at its first call: set here, before the library (and with it the runtime) is loaded, unless the
        # caller's environment already says something.  (A process that initialised HIP earlier -- torch imported first -- has to
        # set it itself: INTEGRATION.md.)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        self.lib = C.CDLL(path)
        for name in EXPORTS:
            getattr(self.lib, name)          # AttributeError if the ABI is incomplete
        self.lib.shasta_mi355x_last_error.restype = C.c_char_p
        self.lib.shasta_mi355x_version.restype = C.c_char_p
        self.lib.shasta_mi355x_create.restype = C.c_void_p
        self.lib.shasta_mi355x_create.argtypes = [C.c_int]
        self.lib.shasta_mi355x_destroy.argtypes = [C.c_void_p]
        self.lib.shasta_mi355x_free.argtypes = [C.c_void_p]

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.shasta_mi355x_last_error().decode()))

    def version(self):
        return self.lib.shasta_mi355x_version().decode()

    def device_count(self):
        return int(self.lib.shasta_mi355x_device_count())

    def environment_warnings(self):
        """Bits: 1 = GPU_MAX_HW_QUEUES unset (the aligner about 15 % slower)."""
        return int(self.lib.shasta_mi355x_environment_warnings())

    # --- one-shot seams -------------------------------------------------------------
    def lowhash0(self, toc, data7, flags, params):
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        read_count = (len(toc) - 1) // 2
        flags = np.zeros(read_count, np.uint8) if flags is None else np.ascontiguousarray(flags, np.uint8)
        stats = np.zeros((read_count, 3), dtype=np.uint64)
        res = abi.LowHash0Result()
        rc = self.lib.shasta_mi355x_lowhash0(
            C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
            abi.as_ptr(flags, C.c_uint8), C.byref(params), abi.as_ptr(stats, C.c_uint64), C.byref(res))
        self._check(rc, "shasta_mi355x_lowhash0")
        out = abi.LowHash0Output(res, stats)
        self.lib.shasta_mi355x_lowhash0_free(C.byref(res))
        return out

    def align4_batch(self, toc, data7, candidates, options, want_ordinals=True):
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        candidates = np.ascontiguousarray(candidates, dtype=abi.PAIR_DTYPE)
        read_count = (len(toc) - 1) // 2
        res = abi.Align4Result()
        rc = self.lib.shasta_mi355x_align4_batch(
            C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
            C.c_uint64(len(candidates)), C.c_void_p(candidates.ctypes.data),
            C.byref(options), C.c_int(1 if want_ordinals else 0), C.byref(res))
        self._check(rc, "shasta_mi355x_align4_batch")
        return abi.Align4Output(res, len(candidates), want_ordinals, free=self.lib.shasta_mi355x_align4_free)

    def align3_batch(self, toc, data7, candidates, options, want_ordinals=True):
        """Align method 3, one-shot on host markers (options: abi.Align3Options)."""
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        candidates = np.ascontiguousarray(candidates, dtype=abi.PAIR_DTYPE)
        read_count = (len(toc) - 1) // 2
        res = abi.Align4Result()
        rc = self.lib.shasta_mi355x_align3_batch(
            C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
            C.c_uint64(len(candidates)), C.c_void_p(candidates.ctypes.data),
            C.byref(options), C.c_int(1 if want_ordinals else 0), C.byref(res))
        self._check(rc, "shasta_mi355x_align3_batch")
        return abi.Align4Output(res, len(candidates), want_ordinals, free=self.lib.shasta_mi355x_align4_free)

    # One-shot calls over several devices (include/shasta_mi355x.h, *_multi): `devices` lists HIP device ids.
    def lowhash0_multi(self, toc, data7, flags, params, devices):
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        read_count = (len(toc) - 1) // 2
        flags = np.zeros(read_count, np.uint8) if flags is None else np.ascontiguousarray(flags, np.uint8)
        stats = np.zeros((read_count, 3), dtype=np.uint64)
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        res = abi.LowHash0Result()
        rc = self.lib.shasta_mi355x_lowhash0_multi(
            C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
            abi.as_ptr(flags, C.c_uint8), C.byref(params), C.c_int(len(dev)), abi.as_ptr(dev, C.c_int), abi.as_ptr(stats, C.c_uint64), C.byref(res))
        self._check(rc, "shasta_mi355x_lowhash0_multi")
        out = abi.LowHash0Output(res, stats)
        self.lib.shasta_mi355x_lowhash0_free(C.byref(res))
        return out

    def _align_batch_multi(self, entry, name, toc, data7, candidates, options, want_ordinals, devices):
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        candidates = np.ascontiguousarray(candidates, dtype=abi.PAIR_DTYPE)
        read_count = (len(toc) - 1) // 2
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        res = abi.Align4Result()
        rc = entry(
            C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64), C.c_void_p(data7.ctypes.data),
            C.c_uint64(len(candidates)), C.c_void_p(candidates.ctypes.data),
            C.byref(options), C.c_int(1 if want_ordinals else 0), C.c_int(len(dev)), abi.as_ptr(dev, C.c_int), C.byref(res))
        self._check(rc, name)
        return abi.Align4Output(res, len(candidates), want_ordinals, free=self.lib.shasta_mi355x_align4_free)

    def align4_batch_multi(self, toc, data7, candidates, options, devices, want_ordinals=True):
        return self._align_batch_multi(self.lib.shasta_mi355x_align4_batch_multi, "shasta_mi355x_align4_batch_multi",
                                       toc, data7, candidates, options, want_ordinals, devices)

    def align3_batch_multi(self, toc, data7, candidates, options, devices, want_ordinals=True):
        return self._align_batch_multi(self.lib.shasta_mi355x_align3_batch_multi, "shasta_mi355x_align3_batch_multi",
                                       toc, data7, candidates, options, want_ordinals, devices)

    def group(self, devices):
        return Group(self, devices)

    def find_markers(self, reads_toc, reads_data, base_counts, k, is_marker, want_packed=True, context=None, flags=None):
        """Marker finding (MarkerFinder).  Reads as Shasta stores them (two bit planes per 64 bases),
        is_marker = one byte per k-mer id.  -> (toc uint64[2R+1], data7 uint8[7*M] or None).  With a
        context the markers stay resident on it, as after set_markers."""
        rt = np.ascontiguousarray(reads_toc, dtype=np.uint64)
        rd = np.ascontiguousarray(reads_data, dtype=np.uint64)
        bc = np.ascontiguousarray(base_counts, dtype=np.uint64)
        im = np.ascontiguousarray(is_marker, dtype=np.uint8)
        fp = abi.as_ptr(np.ascontiguousarray(flags, np.uint8), C.c_uint8) if flags is not None else None
        res = abi.MarkersResult()
        handle = C.c_void_p(context.handle) if context is not None else C.c_void_p(None)
        self._check(self.lib.shasta_mi355x_find_markers(
            handle, C.c_uint64(len(bc)), abi.as_ptr(rt, C.c_uint64), abi.as_ptr(rd, C.c_uint64), abi.as_ptr(bc, C.c_uint64),
            C.c_uint64(k), C.c_void_p(im.ctypes.data), C.c_uint64(1), C.c_uint64(0), fp,
            C.c_int(1 if want_packed else 0), C.byref(res)), "shasta_mi355x_find_markers")
        toc = abi.copy_array(res.markersToc, 2 * len(bc) + 1, "<u8")
        data = abi.copy_array(res.markersData, 7 * int(res.markerCount), "u1") if want_packed else None
        self.lib.shasta_mi355x_find_markers_free(C.byref(res))
        if context is not None:
            context.read_count = len(bc)
        return toc, data

    # --- unit seams -------------------------------------------------------------------
    def hash_windows(self, kmer_ids, m, iteration):
        k = np.ascontiguousarray(kmer_ids, dtype=np.uint32)
        out = np.zeros(max(0, len(k) - m + 1), dtype=np.uint64)
        self._check(self.lib.shasta_mi355x_hash_windows(
            abi.as_ptr(k, C.c_uint32), C.c_uint64(len(k)), C.c_uint64(m), C.c_uint64(iteration),
            abi.as_ptr(out, C.c_uint64)), "shasta_mi355x_hash_windows")
        return out

    def banded_dp(self, k0, k1, band_min, band_max):
        k0 = np.ascontiguousarray(k0, dtype=np.uint32)
        k1 = np.ascontiguousarray(k1, dtype=np.uint32)
        cap = min(len(k0), len(k1)) + 1
        out = np.zeros((cap, 2), dtype=np.uint32)
        count = C.c_uint64()
        score = C.c_int32()
        self._check(self.lib.shasta_mi355x_banded_dp(
            abi.as_ptr(k0, C.c_uint32), C.c_uint32(len(k0)), abi.as_ptr(k1, C.c_uint32), C.c_uint32(len(k1)),
            C.c_int32(band_min), C.c_int32(band_max), abi.as_ptr(out, C.c_uint32), C.c_uint64(cap),
            C.byref(count), C.byref(score)), "shasta_mi355x_banded_dp")
        return out[:count.value].copy(), score.value

    def pair_table(self, pairs, read_count, device=0):
        """computeCandidateTable / computeAlignmentTable: `pairs` is an array of candidates (12-byte records) or of
        AlignmentData rows (64 bytes) -> (toc uint64[2 R + 1], values uint32[4 N])."""
        a = np.ascontiguousarray(pairs)
        stride = a.dtype.itemsize
        toc = np.zeros(2 * int(read_count) + 1, np.uint64)
        values = np.zeros(max(1, 4 * len(a)), np.uint32)
        self._check(self.lib.shasta_mi355x_pair_table(C.c_int(device), C.c_void_p(a.ctypes.data if len(a) else None), C.c_uint64(stride),
                                                      C.c_uint64(len(a)), C.c_uint64(read_count), abi.as_ptr(toc, C.c_uint64),
                                                      abi.as_ptr(values, C.c_uint32)), "shasta_mi355x_pair_table")
        return toc, values[:4 * len(a)]

    def read_graph_keep(self, alignment_data, read_count, max_alignment_count, device=0):
        """createReadGraph's selection -> uint8[N], 1 where the alignment stays."""
        a = np.ascontiguousarray(alignment_data)
        assert a.dtype.itemsize == 64
        keep = np.zeros(max(1, len(a)), np.uint8)
        self._check(self.lib.shasta_mi355x_read_graph_keep(C.c_int(device), C.c_void_p(a.ctypes.data if len(a) else None), C.c_uint64(len(a)),
                                                           C.c_uint64(read_count), C.c_uint32(max_alignment_count), abi.as_ptr(keep, C.c_uint8)),
                    "shasta_mi355x_read_graph_keep")
        return keep[:len(a)]

    def banded_dp_many(self, kmer_ids, begin0, nx, begin1, ny, band_min, band_max, timing=False):
        """K10 on many tasks bundled as in an Align4 batch -> list of (ordinals [n, 2], score) per task.
        timing=True: no ordinals are copied back; returns (counts, scores, seconds[9], cells[8]) instead (eight band classes + the traceback)."""
        k = np.ascontiguousarray(kmer_ids, dtype=np.uint32)
        b0 = np.ascontiguousarray(begin0, np.uint64); b1 = np.ascontiguousarray(begin1, np.uint64)
        n0 = np.ascontiguousarray(nx, np.uint32); n1 = np.ascontiguousarray(ny, np.uint32)
        lo = np.ascontiguousarray(band_min, np.int32); hi = np.ascontiguousarray(band_max, np.int32)
        t = len(b0)
        cap = int(np.minimum(n0, n1).astype(np.uint64).sum()) + 1
        counts = np.zeros(t, np.uint64); scores = np.zeros(t, np.int32)
        seconds = np.zeros(9, np.float64); cells = np.zeros(8, np.uint64)
        if timing:
            self._check(self.lib.shasta_mi355x_banded_dp_many(
                abi.as_ptr(k, C.c_uint32), C.c_uint64(len(k)), C.c_uint64(t),
                abi.as_ptr(b0, C.c_uint64), abi.as_ptr(n0, C.c_uint32), abi.as_ptr(b1, C.c_uint64), abi.as_ptr(n1, C.c_uint32),
                abi.as_ptr(lo, C.c_int32), abi.as_ptr(hi, C.c_int32),
                abi.as_ptr(counts, C.c_uint64), abi.as_ptr(scores, C.c_int32), None, C.c_uint64(0),
                abi.as_ptr(seconds, C.c_double), abi.as_ptr(cells, C.c_uint64)), "shasta_mi355x_banded_dp_many")
            return counts, scores, seconds, cells
        out = np.zeros(2 * cap, np.uint32)
        self._check(self.lib.shasta_mi355x_banded_dp_many(
            abi.as_ptr(k, C.c_uint32), C.c_uint64(len(k)), C.c_uint64(t),
            abi.as_ptr(b0, C.c_uint64), abi.as_ptr(n0, C.c_uint32), abi.as_ptr(b1, C.c_uint64), abi.as_ptr(n1, C.c_uint32),
            abi.as_ptr(lo, C.c_int32), abi.as_ptr(hi, C.c_int32),
            abi.as_ptr(counts, C.c_uint64), abi.as_ptr(scores, C.c_int32), abi.as_ptr(out, C.c_uint32), C.c_uint64(cap),
            None, None), "shasta_mi355x_banded_dp_many")
        ends = np.cumsum(counts).astype(np.int64)
        return [(out[2 * (e - int(c)):2 * e].reshape(-1, 2), int(s)) for e, c, s in zip(ends, counts, scores)]

    def calibrate(self, nbytes, mode):
        self._check(self.lib.shasta_mi355x_calibrate(C.c_uint64(nbytes), C.c_int(mode)), "shasta_mi355x_calibrate")

    def context(self, device=0):
        return Context(self, device)


class Group:
    """Several devices behind one call, markers resident on each (shasta_mi355x_group)."""

    def __init__(self, library, devices):
        self.library = library
        self.lib = library.lib
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        self.lib.shasta_mi355x_group_create.restype = C.c_void_p
        self.handle = self.lib.shasta_mi355x_group_create(C.c_int(len(dev)), abi.as_ptr(dev, C.c_int))
        if not self.handle:
            raise RuntimeError("shasta_mi355x_group_create failed: %s" % self.lib.shasta_mi355x_last_error().decode())
        self.read_count = 0

    def close(self):
        if self.handle:
            self.lib.shasta_mi355x_group_destroy(C.c_void_p(self.handle))
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_kmer_ids(self, toc, kmer_ids, flags=None):
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        kmer = np.ascontiguousarray(kmer_ids, dtype=np.uint32)
        self.read_count = (len(toc) - 1) // 2
        fp = abi.as_ptr(np.ascontiguousarray(flags, np.uint8), C.c_uint8) if flags is not None else None
        self.library._check(self.lib.shasta_mi355x_group_set_kmer_ids(
            C.c_void_p(self.handle), C.c_uint64(self.read_count), abi.as_ptr(toc, C.c_uint64), abi.as_ptr(kmer, C.c_uint32), fp),
            "shasta_mi355x_group_set_kmer_ids")

    def lowhash0(self, params):
        stats = np.zeros((self.read_count, 3), dtype=np.uint64)
        res = abi.LowHash0Result()
        self.library._check(self.lib.shasta_mi355x_group_lowhash0_run(
            C.c_void_p(self.handle), C.byref(params), abi.as_ptr(stats, C.c_uint64), C.byref(res)), "shasta_mi355x_group_lowhash0_run")
        return abi.LowHash0Output(res, stats, free=self.lib.shasta_mi355x_lowhash0_free)       # the candidate list stays in the C buffer

    def align4(self, candidates, options, want_ordinals=False, borrow=False):
        """borrow=True: the arrays of the result are views of memory owned by this group, valid until its next aligner call."""
        candidates = np.ascontiguousarray(candidates, dtype=abi.PAIR_DTYPE)
        res = abi.Align4Result()
        fn = self.lib.shasta_mi355x_group_align4_run_borrowed if borrow else self.lib.shasta_mi355x_group_align4_run
        self.library._check(fn(
            C.c_void_p(self.handle), C.c_uint64(len(candidates)), C.c_void_p(candidates.ctypes.data),
            C.byref(options), C.c_int(1 if want_ordinals else 0), C.byref(res)), "shasta_mi355x_group_align4_run")
        return abi.Align4Output(res, len(candidates), want_ordinals, free=self.lib.shasta_mi355x_align4_free)

    def align3(self, candidates, options, want_ordinals=False, borrow=False):
        candidates = np.ascontiguousarray(candidates, dtype=abi.PAIR_DTYPE)
        res = abi.Align4Result()
        fn = self.lib.shasta_mi355x_group_align3_run_borrowed if borrow else self.lib.shasta_mi355x_group_align3_run
        self.library._check(fn(
            C.c_void_p(self.handle), C.c_uint64(len(candidates)), C.c_void_p(candidates.ctypes.data),
            C.byref(options), C.c_int(1 if want_ordinals else 0), C.byref(res)), "shasta_mi355x_group_align3_run")
        return abi.Align4Output(res, len(candidates), want_ordinals, free=self.lib.shasta_mi355x_align4_free)


class Context:
    """Device-resident markers + the two stages (shasta_mi355x_ctx)."""

    def __init__(self, library, device=0):
        self.library = library
        self.lib = library.lib
        self.handle = self.lib.shasta_mi355x_create(C.c_int(device))
        if not self.handle:
            raise RuntimeError("shasta_mi355x_create failed: %s" % self.lib.shasta_mi355x_last_error().decode())
        self.read_count = 0

    def close(self):
        if self.handle:
            self.lib.shasta_mi355x_destroy(C.c_void_p(self.handle))
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_markers(self, toc, data7, flags=None):
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        data7 = np.ascontiguousarray(data7, dtype=np.uint8)
        self.read_count = (len(toc) - 1) // 2
        fp = abi.as_ptr(np.ascontiguousarray(flags, np.uint8), C.c_uint8) if flags is not None else None
        self.library._check(self.lib.shasta_mi355x_set_markers(
            C.c_void_p(self.handle), C.c_uint64(self.read_count), abi.as_ptr(toc, C.c_uint64),
            C.c_void_p(data7.ctypes.data), fp), "shasta_mi355x_set_markers")

    def set_kmer_ids(self, toc, kmer_ids, flags=None):
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        kmer_ids = np.ascontiguousarray(kmer_ids, dtype=np.uint32)
        self.read_count = (len(toc) - 1) // 2
        fp = abi.as_ptr(np.ascontiguousarray(flags, np.uint8), C.c_uint8) if flags is not None else None
        self.library._check(self.lib.shasta_mi355x_set_kmer_ids(
            C.c_void_p(self.handle), C.c_uint64(self.read_count), abi.as_ptr(toc, C.c_uint64),
            abi.as_ptr(kmer_ids, C.c_uint32), fp), "shasta_mi355x_set_kmer_ids")

    def set_kmer_ids_device(self, toc, kmer_ids_device_ptr, flags=None):
        """Adopts dense kmer ids that already live in this GPU's memory (device pointer as int)."""
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        self.read_count = (len(toc) - 1) // 2
        fp = abi.as_ptr(np.ascontiguousarray(flags, np.uint8), C.c_uint8) if flags is not None else None
        self.library._check(self.lib.shasta_mi355x_set_kmer_ids_device(
            C.c_void_p(self.handle), C.c_uint64(self.read_count), abi.as_ptr(toc, C.c_uint64),
            C.c_void_p(int(kmer_ids_device_ptr)), fp), "shasta_mi355x_set_kmer_ids_device")

    def memcpy(self, dst, src, nbytes, kind):
        """Synchronous copy on the context's stream; kind 0 host->device, 1 device->host, 2 device->device."""
        self.library._check(self.lib.shasta_mi355x_memcpy(
            C.c_void_p(self.handle), C.c_void_p(int(dst)), C.c_void_p(int(src)), C.c_uint64(int(nbytes)), C.c_int(kind)),
            "shasta_mi355x_memcpy")

    # --- LowHash0 in stages (one rank of a job sharded over several GPUs; include/shasta_mi355x.h) ---
    def lh_begin(self, params, rank, world, boundaries):
        b = np.ascontiguousarray(boundaries, dtype=np.uint64)
        assert len(b) == world + 1
        log2 = C.c_uint32()
        self.library._check(self.lib.shasta_mi355x_lh_begin(
            C.c_void_p(self.handle), C.byref(params), C.c_int(rank), C.c_int(world), abi.as_ptr(b, C.c_uint64),
            C.byref(log2)), "shasta_mi355x_lh_begin")
        self._world = world
        self._planned_iterations = int(params.minHashIterationCount)
        return int(log2.value)

    def lh_one_pass_fits(self):
        fits = C.c_int(0)
        self.library._check(self.lib.shasta_mi355x_lh_one_pass_fits(C.c_void_p(self.handle), C.byref(fits)), "shasta_mi355x_lh_one_pass_fits")
        return bool(fits.value)

    def lh_hash(self, iteration):
        """-> (send offsets uint64[world+1], device pointer of keys u32, device pointer of vals u64)."""
        offsets = np.zeros(self._world + 1, dtype=np.uint64)
        keys, vals = C.c_void_p(), C.c_void_p()
        self.library._check(self.lib.shasta_mi355x_lh_hash(
            C.c_void_p(self.handle), C.c_uint64(iteration), abi.as_ptr(offsets, C.c_uint64),
            C.byref(keys), C.byref(vals)), "shasta_mi355x_lh_hash")
        return offsets, keys.value or 0, vals.value or 0

    def lh_buckets(self, keys_ptr, vals_ptr, n):
        """-> (send offsets, pair keys ptr (u64, sorted), bucketsUsed, size histogram uint64[2048], overflow sizes)."""
        offsets = np.zeros(self._world + 1, dtype=np.uint64)
        pair_keys = C.c_void_p()
        used, overflow_count = C.c_uint64(), C.c_uint64()
        hist = np.zeros(2048, dtype=np.uint64)
        overflow = np.zeros(1 << 20, dtype=np.uint32)
        self.library._check(self.lib.shasta_mi355x_lh_buckets(
            C.c_void_p(self.handle), C.c_void_p(int(keys_ptr)), C.c_void_p(int(vals_ptr)), C.c_uint64(int(n)),
            abi.as_ptr(offsets, C.c_uint64), C.byref(pair_keys), C.byref(used),
            abi.as_ptr(hist, C.c_uint64), abi.as_ptr(overflow, C.c_uint32), C.c_uint64(len(overflow)),
            C.byref(overflow_count)), "shasta_mi355x_lh_buckets")
        return offsets, pair_keys.value or 0, int(used.value), hist, overflow[:overflow_count.value].copy()

    def lh_merge(self, pair_keys_ptr, n, evaluate_now=False):
        """Appends this iteration's keys; with evaluate_now -> this rank's (high frequency, total) of the iteration, else zeros."""
        high, total = C.c_uint64(), C.c_uint64()
        self.library._check(self.lib.shasta_mi355x_lh_merge(
            C.c_void_p(self.handle), C.c_void_p(int(pair_keys_ptr)), C.c_uint64(int(n)), C.c_int(1 if evaluate_now else 0),
            C.byref(high), C.byref(total)), "shasta_mi355x_lh_merge")
        return int(high.value), int(total.value)

    # The same job with all iterations in one pass (fixed minHashIterationCount): one call of each per job.
    def lh_hash_all(self):
        """-> (send offsets uint64[world+1], device pointer of keys u64 (owner << 56 | iteration << 32 | bucket id), of vals u64)."""
        offsets = np.zeros(self._world + 1, dtype=np.uint64)
        keys, vals = C.c_void_p(), C.c_void_p()
        self.library._check(self.lib.shasta_mi355x_lh_hash_all(
            C.c_void_p(self.handle), abi.as_ptr(offsets, C.c_uint64), C.byref(keys), C.byref(vals)), "shasta_mi355x_lh_hash_all")
        return offsets, keys.value or 0, vals.value or 0

    def lh_buckets_all(self, keys_ptr, vals_ptr, n):
        """-> (send offsets, pair keys ptr (u64, sorted), their iteration tags ptr (u32), bucketsUsed uint64[iterations],
        size histograms uint64[iterations, 2048], overflow entries uint64 (iteration << 32 | size))."""
        iterations = self._planned_iterations
        offsets = np.zeros(self._world + 1, dtype=np.uint64)
        pair_keys, pair_tags = C.c_void_p(), C.c_void_p()
        overflow_count = C.c_uint64()
        used = np.zeros(max(1, iterations), dtype=np.uint64)
        hist = np.zeros((max(1, iterations), 2048), dtype=np.uint64)
        overflow = np.zeros(1 << 20, dtype=np.uint64)
        self.library._check(self.lib.shasta_mi355x_lh_buckets_all(
            C.c_void_p(self.handle), C.c_void_p(int(keys_ptr)), C.c_void_p(int(vals_ptr)), C.c_uint64(int(n)),
            abi.as_ptr(offsets, C.c_uint64), C.byref(pair_keys), C.byref(pair_tags), C.c_uint64(max(1, iterations)), abi.as_ptr(used, C.c_uint64),
            abi.as_ptr(hist, C.c_uint64), abi.as_ptr(overflow, C.c_uint64), C.c_uint64(len(overflow)),
            C.byref(overflow_count)), "shasta_mi355x_lh_buckets_all")
        return offsets, pair_keys.value or 0, pair_tags.value or 0, used[:iterations], hist[:iterations], overflow[:overflow_count.value].copy()

    def lh_merge_all(self, pair_keys_ptr, pair_tags_ptr, n):
        self.library._check(self.lib.shasta_mi355x_lh_merge_all(
            C.c_void_p(self.handle), C.c_void_p(int(pair_keys_ptr)), C.c_void_p(int(pair_tags_ptr)), C.c_uint64(int(n))), "shasta_mi355x_lh_merge_all")

    def lh_finish(self, max_iterations=1 << 16):
        """-> (this rank's candidates, its partial statistics uint64[R,3], its share of high frequency / total per iteration)."""
        stats = np.zeros((self.read_count, 3), dtype=np.uint64)
        cand = C.POINTER(abi.OrientedReadPair)()
        count, iterations = C.c_uint64(), C.c_uint64()
        high = np.zeros(max_iterations, dtype=np.uint64)
        total = np.zeros(max_iterations, dtype=np.uint64)
        self.library._check(self.lib.shasta_mi355x_lh_finish(
            C.c_void_p(self.handle), abi.as_ptr(stats, C.c_uint64), C.byref(cand), C.byref(count),
            abi.as_ptr(high, C.c_uint64), abi.as_ptr(total, C.c_uint64), C.c_uint64(max_iterations), C.byref(iterations)),
            "shasta_mi355x_lh_finish")
        out = abi.copy_array(cand, int(count.value), abi.PAIR_DTYPE)
        self.lib.shasta_mi355x_free(cand)
        k = int(iterations.value)
        return out, stats, high[:k].copy(), total[:k].copy()

    def lh_finish_on_device(self, max_iterations=1 << 16):
        """-> (device address of this rank's candidates, their number, statistics, high frequency / total per iteration): the
        candidates stay in device memory, valid until this context's next LowHash0 job."""
        stats = np.zeros((self.read_count, 3), dtype=np.uint64)
        cand, count, iterations = C.c_void_p(), C.c_uint64(), C.c_uint64()
        high = np.zeros(max_iterations, dtype=np.uint64)
        total = np.zeros(max_iterations, dtype=np.uint64)
        self.library._check(self.lib.shasta_mi355x_lh_finish_on_device(
            C.c_void_p(self.handle), abi.as_ptr(stats, C.c_uint64), C.byref(cand), C.byref(count),
            abi.as_ptr(high, C.c_uint64), abi.as_ptr(total, C.c_uint64), C.c_uint64(max_iterations), C.byref(iterations)),
            "shasta_mi355x_lh_finish_on_device")
        k = int(iterations.value)
        return cand.value or 0, int(count.value), stats, high[:k].copy(), total[:k].copy()

    def lowhash0(self, params):
        stats = np.zeros((self.read_count, 3), dtype=np.uint64)
        res = abi.LowHash0Result()
        self.library._check(self.lib.shasta_mi355x_lowhash0_run(
            C.c_void_p(self.handle), C.byref(params), abi.as_ptr(stats, C.c_uint64), C.byref(res)),
            "shasta_mi355x_lowhash0_run")
        return abi.LowHash0Output(res, stats, free=self.lib.shasta_mi355x_lowhash0_free)       # the candidate list stays in the C buffer

    def align4(self, candidates, options, want_ordinals=False, borrow=False):
        """borrow=True: the arrays of the result are views of memory owned by this context, valid
        until the next borrowed call (align4_run_borrowed)."""
        candidates = np.ascontiguousarray(candidates, dtype=abi.PAIR_DTYPE)
        res = abi.Align4Result()
        entry = self.lib.shasta_mi355x_align4_run_borrowed if borrow else self.lib.shasta_mi355x_align4_run
        self.library._check(entry(
            C.c_void_p(self.handle), C.c_uint64(len(candidates)), C.c_void_p(candidates.ctypes.data),
            C.byref(options), C.c_int(1 if want_ordinals else 0), C.byref(res)), "shasta_mi355x_align4_run")
        return abi.Align4Output(res, len(candidates), want_ordinals, free=self.lib.shasta_mi355x_align4_free)

    def align3(self, candidates, options, want_ordinals=False, borrow=False):
        """Align method 3 on the resident markers (options: abi.Align3Options)."""
        candidates = np.ascontiguousarray(candidates, dtype=abi.PAIR_DTYPE)
        res = abi.Align4Result()
        self.library._check(self.lib.shasta_mi355x_align3_run(
            C.c_void_p(self.handle), C.c_uint64(len(candidates)), C.c_void_p(candidates.ctypes.data),
            C.byref(options), C.c_int(1 if want_ordinals else 0), C.c_int(1 if borrow else 0), C.byref(res)),
            "shasta_mi355x_align3_run")
        return abi.Align4Output(res, len(candidates), want_ordinals, free=self.lib.shasta_mi355x_align4_free)

    def alignment_table(self, copy=True):
        """Assembler::computeAlignmentTable for the alignments of this context's last borrowed aligner call ->
        (toc uint64[2 R + 1], values uint32[4 N]); copy=False: views of the context's page-locked arrays, valid until its
        next aligner or table call."""
        toc, values, n = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)(), C.c_uint64(0)
        self.library._check(self.lib.shasta_mi355x_alignment_table(C.c_void_p(self.handle), C.byref(toc), C.byref(values), C.byref(n)),
                            "shasta_mi355x_alignment_table")
        t = np.ctypeslib.as_array(toc, shape=(2 * self.read_count + 1,))
        v = np.ctypeslib.as_array(values, shape=(max(1, n.value),))[:n.value]
        return (t.copy(), v.copy()) if copy else (t, v)

    def palindromic_screen(self, delta_threshold):
        """Per read: an upper bound on the near-diagonal marker count of its method-0 self-alignment."""
        bound = np.zeros(self.read_count, dtype=np.uint32)
        self.library._check(self.lib.shasta_mi355x_palindromic_screen(
            C.c_void_p(self.handle), C.c_uint64(delta_threshold), abi.as_ptr(bound, C.c_uint32)),
            "shasta_mi355x_palindromic_screen")
        return bound

    def kernel_table(self):
        """{kernel name: {"seconds", "launches", "bytes", "work"}} accumulated on this context since the last reset."""
        rows = (abi.KernelStat * 128)()
        count = C.c_uint64(0)
        self.library._check(self.lib.shasta_mi355x_kernel_table(C.c_void_p(self.handle), rows, C.c_uint64(128), C.byref(count)),
                            "shasta_mi355x_kernel_table")
        return {rows[k].name.decode(): {"seconds": rows[k].seconds, "launches": int(rows[k].launches),
                                        "bytes": int(rows[k].algorithmicBytes), "work": int(rows[k].work)}
                for k in range(min(128, count.value))}

    def kernel_table_reset(self):
        self.library._check(self.lib.shasta_mi355x_kernel_table_reset(C.c_void_p(self.handle)), "shasta_mi355x_kernel_table_reset")


_cached = None


def load():
    """The product library.  SHASTA_MI355X_LIBRARY names another in-tree hipcc build of the SAME sources
    (other compile-time options, for A/B timing on the GPU box); it is still the HIP library: no fallback."""
    global _cached
    if _cached is None:
        _cached = Library(os.environ.get("SHASTA_MI355X_LIBRARY") or SO_PATH)
    return _cached
