"""ctypes mirror of include/shasta_mi355x.h (PODs only) and numpy marshalling helpers.

Every structure here is declared field for field as in the C header; the sizes
are asserted against the reference's struct sizes (SURVEY.md Appendix B:
OrientedReadPair 12, AlignmentInfo 52, AlignmentData 64).
"""
import ctypes as C

import numpy as np


class OrientedReadPair(C.Structure):
    _fields_ = [("readIds", C.c_uint32 * 2), ("isSameStrand", C.c_uint8), ("pad", C.c_uint8 * 3)]


class _InfoData(C.Structure):
    _fields_ = [("markerCount", C.c_uint32), ("firstOrdinal", C.c_uint32), ("lastOrdinal", C.c_uint32)]


class AlignmentInfo(C.Structure):
    _fields_ = [
        ("data", _InfoData * 2),
        ("markerCount", C.c_uint32),
        ("minOrdinalOffset", C.c_int32),
        ("maxOrdinalOffset", C.c_int32),
        ("averageOrdinalOffset", C.c_int32),
        ("maxSkip", C.c_uint32),
        ("maxDrift", C.c_uint32),
        ("isInReadGraph", C.c_uint8),
        ("pad", C.c_uint8 * 3),
    ]


class AlignmentData(C.Structure):
    _fields_ = [("pair", OrientedReadPair), ("info", AlignmentInfo)]


class LowHash0Params(C.Structure):
    _fields_ = [
        ("m", C.c_uint64),
        ("hashFraction", C.c_double),
        ("minHashIterationCount", C.c_uint64),
        ("alignmentCandidatesPerRead", C.c_double),
        ("log2MinHashBucketCount", C.c_uint64),
        ("minBucketSize", C.c_uint64),
        ("maxBucketSize", C.c_uint64),
        ("minFrequency", C.c_uint64),
    ]


class LowHash0Result(C.Structure):
    _fields_ = [
        ("candidateCount", C.c_uint64),
        ("candidates", C.POINTER(OrientedReadPair)),
        ("log2BucketCount", C.c_uint32),
        ("iterationCount", C.c_uint32),
        ("highFrequency", C.POINTER(C.c_uint64)),
        ("total", C.POINTER(C.c_uint64)),
        ("histogramRowCount", C.c_uint64),
        ("histogram", C.POINTER(C.c_uint64)),
        ("seconds", C.c_double),
        ("deviceSeconds", C.c_double),
    ]


class Align4Options(C.Structure):
    _fields_ = [
        ("deltaX", C.c_uint64),
        ("deltaY", C.c_uint64),
        ("minEntryCountPerCell", C.c_uint64),
        ("maxDistanceFromBoundary", C.c_uint64),
        ("minAlignedMarkerCount", C.c_uint64),
        ("minAlignedFraction", C.c_double),
        ("maxSkip", C.c_uint64),
        ("maxDrift", C.c_uint64),
        ("maxTrim", C.c_uint64),
        ("maxBand", C.c_uint64),
        ("matchScore", C.c_int64),
        ("mismatchScore", C.c_int64),
        ("gapScore", C.c_int64),
        ("suppressContainments", C.c_uint8),
        ("pad", C.c_uint8 * 7),
    ]


class Align3Options(C.Structure):
    _fields_ = [
        ("matchScore", C.c_int64),
        ("mismatchScore", C.c_int64),
        ("gapScore", C.c_int64),
        ("downsamplingFactor", C.c_double),
        ("bandExtend", C.c_int64),
        ("maxBand", C.c_int64),
        ("k", C.c_uint64),
        ("minAlignedMarkerCount", C.c_uint64),
        ("minAlignedFraction", C.c_double),
        ("maxSkip", C.c_uint64),
        ("maxDrift", C.c_uint64),
        ("maxTrim", C.c_uint64),
        ("suppressContainments", C.c_uint8),
        ("pad", C.c_uint8 * 7),
    ]


class Align4Result(C.Structure):
    _fields_ = [
        ("alignmentCount", C.c_uint64),
        ("alignmentData", C.POINTER(AlignmentData)),
        ("compressedToc", C.POINTER(C.c_uint64)),
        ("compressedData", C.POINTER(C.c_uint8)),
        ("status", C.POINTER(C.c_uint8)),
        ("ordinalsToc", C.POINTER(C.c_uint64)),
        ("ordinals", C.POINTER(C.c_uint32)),
        ("dpCellCount", C.c_uint64),
        ("kmerIdBytes", C.c_uint64),
        ("alignedBytes", C.c_uint64),
        ("seconds", C.c_double),
        ("deviceSeconds", C.c_double),
        ("owner", C.c_void_p),
    ]


class MarkersResult(C.Structure):
    _fields_ = [
        ("markerCount", C.c_uint64),
        ("markersToc", C.POINTER(C.c_uint64)),
        ("markersData", C.POINTER(C.c_uint8)),
        ("seconds", C.c_double),
        ("deviceSeconds", C.c_double),
    ]


class KernelStat(C.Structure):
    _fields_ = [
        ("name", C.c_char * 64),
        ("seconds", C.c_double),
        ("launches", C.c_uint64),
        ("algorithmicBytes", C.c_uint64),
        ("work", C.c_uint64),
    ]


assert C.sizeof(OrientedReadPair) == 12
assert C.sizeof(AlignmentInfo) == 52
assert C.sizeof(AlignmentData) == 64
assert C.sizeof(Align4Options) == 112
assert C.sizeof(Align3Options) == 104

SHASTA_ALIGN_STORED = 0
SHASTA_ALIGN_REJECTED = 1
SHASTA_ALIGN_EMPTY = 2
SHASTA_ALIGN_SKIPPED = 3
SHASTA_ALIGN_TIE_FLAG = 0x80

PAIR_DTYPE = np.dtype(
    {"names": ["readId0", "readId1", "isSameStrand"],
     "formats": ["<u4", "<u4", "u1"], "offsets": [0, 4, 8], "itemsize": 12})

INFO_FIELDS = ["markerCount0", "firstOrdinal0", "lastOrdinal0",
               "markerCount1", "firstOrdinal1", "lastOrdinal1",
               "markerCount", "minOrdinalOffset", "maxOrdinalOffset",
               "averageOrdinalOffset", "maxSkip", "maxDrift"]
ALIGNMENT_DATA_DTYPE = np.dtype(
    {"names": ["readId0", "readId1", "isSameStrand"] + INFO_FIELDS,
     "formats": ["<u4", "<u4", "u1"] + ["<u4"] * 7 + ["<i4"] * 3 + ["<u4"] * 2,
     "offsets": [0, 4, 8] + [12 + 4 * i for i in range(12)], "itemsize": 64})


def default_lowhash0_params(**kw):
    """MinHash defaults of src/AssemblerOptions.cpp:327-371."""
    p = LowHash0Params(m=4, hashFraction=0.01, minHashIterationCount=10,
                       alignmentCandidatesPerRead=20.0, log2MinHashBucketCount=0,
                       minBucketSize=0, maxBucketSize=10, minFrequency=2)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_align4_options(**kw):
    """Align.* defaults of src/AssemblerOptions.cpp:380-489 (align4.* 200/10/10/100)."""
    o = Align4Options(deltaX=200, deltaY=10, minEntryCountPerCell=10, maxDistanceFromBoundary=100,
                      minAlignedMarkerCount=100, minAlignedFraction=0.0, maxSkip=30, maxDrift=30,
                      maxTrim=30, maxBand=1000, matchScore=6, mismatchScore=-1, gapScore=-1,
                      suppressContainments=0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def default_align3_options(**kw):
    """Align.* defaults of src/AssemblerOptions.cpp:380-449 for alignMethod 3 (k: Kmers.k default 10)."""
    o = Align3Options(matchScore=6, mismatchScore=-1, gapScore=-1, downsamplingFactor=0.1, bandExtend=10,
                      maxBand=1000, k=10, minAlignedMarkerCount=100, minAlignedFraction=0.0, maxSkip=30,
                      maxDrift=30, maxTrim=30, suppressContainments=0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def as_ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def copy_array(ptr, n, dtype):
    """Copy n items of numpy dtype from a ctypes pointer (None/NULL -> empty)."""
    dtype = np.dtype(dtype)
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    buf = C.cast(ptr, C.POINTER(C.c_uint8 * (n * dtype.itemsize))).contents
    # Copied as bytes: numpy copies a record type with padding (the 12-byte pair) field by field -- 9 ms for 2 M candidates
    # against 1 ms for their 24 MB.
    return np.frombuffer(buf, dtype=np.uint8, count=n * dtype.itemsize).copy().view(dtype)


def view_array(ptr, n, dtype, owner):
    """Zero-copy numpy view of n items behind a ctypes pointer; `owner` keeps the C buffers alive."""
    dtype = np.dtype(dtype)
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    buf = C.cast(ptr, C.POINTER(C.c_uint8 * (n * dtype.itemsize))).contents
    buf._owner = owner                           # every array and slice made from the view keeps the C buffers alive
    a = np.frombuffer(buf, dtype=dtype, count=n)
    a.flags.writeable = False
    return a


class _ResultOwner:
    """Owns a C result struct; the library's free function runs when the last view is gone."""

    def __init__(self, res, free):
        self.res, self.free = res, free

    def __del__(self):
        try:
            self.free(C.byref(self.res))
        except Exception:
            pass


class LowHash0Output:
    """shasta_lowhash0_result + the statistics table.  With `free` (the library's shasta_mi355x_lowhash0_free) the candidate
    list is a zero-copy, read-only view of the C buffer, released when this object goes away; without it, a copy."""

    def __init__(self, res, stats, free=None):
        if free is not None:
            self._owner = _ResultOwner(res, free)
            self.candidates = view_array(res.candidates, res.candidateCount, PAIR_DTYPE, self._owner)
        else:
            self.candidates = copy_array(res.candidates, res.candidateCount, PAIR_DTYPE)
        self.log2_bucket_count = int(res.log2BucketCount)
        self.high_frequency = copy_array(res.highFrequency, res.iterationCount, "<u8")
        self.total = copy_array(res.total, res.iterationCount, "<u8")
        self.histogram = copy_array(res.histogram, 3 * res.histogramRowCount, "<u8").reshape(-1, 3)
        self.statistics = stats
        self.seconds = float(res.seconds)
        self.device_seconds = float(res.deviceSeconds)

    def candidate_tuples(self):
        c = self.candidates
        return np.stack([c["readId0"], c["readId1"], c["isSameStrand"].astype(np.uint32)], axis=1)


class Align4Output:
    def __init__(self, res, candidate_count, want_ordinals, free=None):
        """With `free` (the library's *_free function) the arrays are zero-copy views of the C
        buffers, released when this object goes away; without it they are copies."""
        if free is not None:
            self._owner = _ResultOwner(res, free)
            get = lambda ptr, n, dtype: view_array(ptr, n, dtype, self._owner)
        else:
            get = copy_array
        n = int(res.alignmentCount)
        self.alignment_data = get(res.alignmentData, n, ALIGNMENT_DATA_DTYPE)
        self.compressed_toc = get(res.compressedToc, n + 1, "<u8")
        nbytes = int(self.compressed_toc[-1]) if n + 1 > 0 and len(self.compressed_toc) else 0
        self.compressed_data = get(res.compressedData, nbytes, "u1")
        self.status = get(res.status, candidate_count, "u1")
        if want_ordinals and res.ordinalsToc:
            self.ordinals_toc = get(res.ordinalsToc, candidate_count + 1, "<u8")
            self.ordinals = get(res.ordinals, 2 * int(self.ordinals_toc[-1]), "<u4").reshape(-1, 2)
        else:
            self.ordinals_toc = None
            self.ordinals = None
        self.dp_cell_count = int(res.dpCellCount)
        self.kmer_id_bytes = int(res.kmerIdBytes)
        self.aligned_bytes = int(res.alignedBytes)
        self.seconds = float(res.seconds)
        self.device_seconds = float(res.deviceSeconds)

    def info_table(self):
        a = self.alignment_data
        return np.stack([a[f].astype(np.int64) for f in
                         ["readId0", "readId1", "isSameStrand"] + INFO_FIELDS], axis=1)

    def ordinals_of(self, i):
        return self.ordinals[int(self.ordinals_toc[i]):int(self.ordinals_toc[i + 1])]

    def per_candidate(self, keep=None):
        """[(candidate index, status, ordinals bytes or None, AlignmentInfo row bytes or None, compressed blob or None)]
        for the candidates with keep[i] (all when keep is None): two results can be compared candidate by candidate."""
        stored = (self.status & 0x7f) == SHASTA_ALIGN_STORED
        rows = np.cumsum(stored) - 1
        info = self.info_table()
        which = np.arange(len(self.status)) if keep is None else np.flatnonzero(keep)
        items = []
        for i in which:
            i = int(i)
            ords = None if self.ordinals_toc is None else self.ordinals_of(i).tobytes()
            if not stored[i]:
                items.append((i, int(self.status[i]), ords, None, None))
                continue
            r = int(rows[i])
            blob = self.compressed_data[int(self.compressed_toc[r]):int(self.compressed_toc[r + 1])].tobytes()
            items.append((i, int(self.status[i]), ords, info[r].tobytes(), blob))
        return items


def make_pairs(readId0, readId1, isSameStrand):
    n = len(readId0)
    a = np.zeros(n, dtype=PAIR_DTYPE)
    a["readId0"] = readId0
    a["readId1"] = readId1
    a["isSameStrand"] = isSameStrand
    return a
