"""The stage-level Python surface of the reference for this path: `shasta.Assembler` and
`shasta.AlignOptions` (pybind11, src/PythonModule.cpp:85-107,135-345) with the same method and
attribute names, over an existing Data/ directory, for the two hot functions and the table step
between them.  The reference's stage scripts (scripts/FindAlignmentCandidatesLowHash0.py,
scripts/ComputeAlignments.py) run unchanged with `import shasta_amd.assembler as shasta`.

The work happens in libshasta_mi355x_host.so (C++ host layer, shasta_amd/host/) which calls the GPU
library through its C ABI; there is no CPU fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SHASTA_MI355X_HOST_LIBRARY: a developer switch for the stage scripts (tests point it at the emulated twin of the host library).
HOST_SO = os.environ.get("SHASTA_MI355X_HOST_LIBRARY") or os.path.join(HERE, "_build", "libshasta_mi355x_host.so")


class _HostAlignOptions(C.Structure):
    _fields_ = [
        ("alignMethod", C.c_int64),
        ("maxSkip", C.c_uint64), ("maxDrift", C.c_uint64), ("maxTrim", C.c_uint64), ("minAlignedMarkerCount", C.c_uint64),
        ("minAlignedFraction", C.c_double),
        ("matchScore", C.c_int64), ("mismatchScore", C.c_int64), ("gapScore", C.c_int64), ("maxBand", C.c_int64),
        ("suppressContainments", C.c_uint64),
        ("align4DeltaX", C.c_uint64), ("align4DeltaY", C.c_uint64),
        ("align4MinEntryCountPerCell", C.c_uint64), ("align4MaxDistanceFromBoundary", C.c_uint64),
        ("downsamplingFactor", C.c_double), ("bandExtend", C.c_int64),
    ]


class AlignOptions:
    """shasta.AlignOptions: same attributes (src/PythonModule.cpp:85-107), defaults of
    src/AssemblerOptions.cpp:380-489 except alignMethod, which stays 4 here (the reference's default is 3;
    this library implements both, method 3 has not run on the MI355X yet)."""

    def __init__(self):
        self.alignMethod = 4
        self.maxSkip = 30
        self.maxDrift = 30
        self.maxTrim = 30
        self.maxMarkerFrequency = 10                 # methods 0/1/3 only
        self.minAlignedMarkerCount = 100
        self.minAlignedFraction = 0.0
        self.matchScore = 6
        self.mismatchScore = -1
        self.gapScore = -1
        self.downsamplingFactor = 0.1                # method 3 only
        self.bandExtend = 10                         # method 3 only
        self.maxBand = 1000
        self.sameChannelReadAlignmentSuppressDeltaThreshold = 0
        self.suppressContainments = False
        self.align4DeltaX = 200
        self.align4DeltaY = 10
        self.align4MinEntryCountPerCell = 10
        self.align4MaxDistanceFromBoundary = 100


class Assembler:
    """shasta.Assembler for the overlap-detection stages.  Construct it on a run's Data/ directory
    (the reference's default largeDataFileNamePrefix is "Data/")."""

    def __init__(self, largeDataFileNamePrefix="Data/", createNew=False, readRepresentation=1, largeDataPageSize=4096,
                 hostLibrary=HOST_SO):
        if createNew:
            raise RuntimeError("shasta_amd.Assembler works on an existing Data/ directory (createNew is not supported).")
        if not os.path.exists(hostLibrary):
            raise RuntimeError("%s is missing: build with `python -c 'import __graft_entry__ as g; g.build()'`. "
                               "There is no CPU fallback." % hostLibrary)
        self._lib = C.CDLL(hostLibrary)
        self._lib.shasta_mi355x_host_last_error.restype = C.c_char_p
        self._data = largeDataFileNamePrefix.rstrip("/") or "."
        self._page = int(largeDataPageSize)

    def _check(self, rc):
        if rc:
            raise RuntimeError(self._lib.shasta_mi355x_host_last_error().decode())

    def _require(self, *names):
        for name in names:
            if not os.path.exists(os.path.join(self._data, name)):
                raise RuntimeError("Error accessing %s: the file could not be opened." % os.path.join(self._data, name))

    # The reference's access* calls map the files; here they check that the files exist.
    def accessKmers(self):
        pass      # only its size is used (k, for alignMethod 3); LowHash0 and method 4 never read it

    def accessMarkers(self):
        self._require("Markers.toc", "Markers.data", "ReadFlags")

    def accessAlignmentCandidates(self):
        self._require("AlignmentCandidates")

    def findMarkers(self, threadCount=0):
        """shasta.Assembler.findMarkers (src/PythonModule.cpp; src/AssemblerMarkers.cpp:11-24)."""
        self._require("Reads-Bases.toc", "Reads-Bases.data", "Reads-BaseCount", "Kmers")
        self._check(self._lib.shasta_mi355x_host_find_markers(self._data.encode(), C.c_uint64(threadCount), C.c_uint64(self._page)))

    def flagPalindromicReads(self, maxSkip, maxDrift, maxMarkerFrequency, alignedFractionThreshold,
                             nearDiagonalFractionThreshold, deltaThreshold, threadCount=0):
        """shasta.Assembler.flagPalindromicReads (src/PythonModule.cpp:251-259; src/AssemblerAlign.cpp:652-698):
        sets the isPalindromic bit of Data/ReadFlags.  Returns (reads, reads settled by the device screen, flagged)."""
        self._require("Markers.toc", "Markers.data", "ReadFlags")
        counts = (C.c_uint64 * 3)()
        self._check(self._lib.shasta_mi355x_host_flag_palindromic_reads(
            self._data.encode(), C.c_uint32(maxSkip), C.c_uint32(maxDrift), C.c_uint32(maxMarkerFrequency),
            C.c_double(alignedFractionThreshold), C.c_double(nearDiagonalFractionThreshold), C.c_uint32(deltaThreshold),
            C.c_uint64(threadCount), counts))
        return int(counts[0]), int(counts[1]), int(counts[2])

    def findAlignmentCandidatesLowHash0(self, m, hashFraction, minHashIterationCount, alignmentCandidatesPerRead,
                                        minBucketSize, maxBucketSize, minFrequency, log2MinHashBucketCount=0, threadCount=0):
        self._check(self._lib.shasta_mi355x_host_find_alignment_candidates_lowhash0(
            self._data.encode(), C.c_uint64(m), C.c_double(hashFraction), C.c_uint64(minHashIterationCount),
            C.c_double(alignmentCandidatesPerRead), C.c_uint64(log2MinHashBucketCount), C.c_uint64(minBucketSize),
            C.c_uint64(maxBucketSize), C.c_uint64(minFrequency), C.c_uint64(threadCount), C.c_uint64(self._page)))

    def suppressAlignmentCandidates(self, delta, threadCount=0):
        """shasta.Assembler.suppressAlignmentCandidates (src/AssemblerAlign.cpp:1168-1240): host step between the seams."""
        self._require("ReadNames.toc", "ReadNames.data", "ReadMetaData.toc", "ReadMetaData.data", "AlignmentCandidates")
        n = C.c_uint64()
        self._check(self._lib.shasta_mi355x_host_suppress_alignment_candidates(self._data.encode(), C.c_uint64(delta), C.c_uint64(threadCount), C.byref(n)))
        return int(n.value)

    def computeCandidateTable(self):
        self._check(self._lib.shasta_mi355x_host_compute_candidate_table(self._data.encode(), C.c_uint64(self._page)))

    def accessAlignmentData(self):
        self._require("AlignmentData", "AlignmentTable.toc", "AlignmentTable.data")

    def createReadGraph(self, maxAlignmentCount, maxTrim):
        """ReadGraph.creationMethod 0 (src/AssemblerReadGraph.cpp:35-104): host work on the stored alignments."""
        self._check(self._lib.shasta_mi355x_host_create_read_graph(
            self._data.encode(), C.c_uint32(maxAlignmentCount), C.c_uint32(maxTrim), C.c_uint64(self._page)))

    def computeAlignments(self, alignOptions, threadCount=0):
        o = _HostAlignOptions(
            alignMethod=int(alignOptions.alignMethod), maxSkip=int(alignOptions.maxSkip), maxDrift=int(alignOptions.maxDrift),
            maxTrim=int(alignOptions.maxTrim), minAlignedMarkerCount=int(alignOptions.minAlignedMarkerCount),
            minAlignedFraction=float(alignOptions.minAlignedFraction), matchScore=int(alignOptions.matchScore),
            mismatchScore=int(alignOptions.mismatchScore), gapScore=int(alignOptions.gapScore), maxBand=int(alignOptions.maxBand),
            suppressContainments=1 if alignOptions.suppressContainments else 0,
            align4DeltaX=int(alignOptions.align4DeltaX), align4DeltaY=int(alignOptions.align4DeltaY),
            align4MinEntryCountPerCell=int(alignOptions.align4MinEntryCountPerCell),
            align4MaxDistanceFromBoundary=int(alignOptions.align4MaxDistanceFromBoundary),
            downsamplingFactor=float(alignOptions.downsamplingFactor), bandExtend=int(alignOptions.bandExtend))
        self._check(self._lib.shasta_mi355x_host_compute_alignments(self._data.encode(), C.byref(o), C.c_uint64(threadCount),
                                                                    C.c_uint64(self._page)))


def suppress_candidates_in_memory(candidates, meta_data, delta, hostLibrary=None):
    """Assembler::suppressAlignmentCandidates (src/AssemblerAlign.cpp:1168-1240) on arrays in memory -- the step the assemble path
    runs between the two seams (srcMain/main.cpp:697-702): `candidates` = the first seam's OrientedReadPair array, `meta_data` =
    (toc uint64[R + 1], bytes) in the layout of Data/ReadMetaData.  Returns the candidates that stay (a prefix of a copy)."""
    import numpy as np
    lib = C.CDLL(hostLibrary or HOST_SO)
    lib.shasta_mi355x_host_last_error.restype = C.c_char_p
    toc, data = meta_data
    toc = np.ascontiguousarray(toc, dtype=np.uint64)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.array(candidates, copy=True)
    kept = C.c_uint64()
    rc = lib.shasta_mi355x_host_suppress_candidates_in_memory(
        toc.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p), C.c_uint64(len(toc) - 1),
        out.ctypes.data_as(C.c_void_p), C.c_uint64(len(out)), C.c_uint64(delta), C.byref(kept))
    if rc != 0:
        raise RuntimeError(lib.shasta_mi355x_host_last_error().decode())
    return out[:int(kept.value)]


class CandidateSuppression:
    """The same step with the reads' meta data parsed ONCE (a run's meta data does not change between calls): keys per read at
    construction, then `apply(candidates)` -> the candidates that stay, per call a pass of integer comparisons on a few host threads."""

    def __init__(self, meta_data, delta, hostLibrary=None, threads=8):
        import numpy as np
        self._lib = C.CDLL(hostLibrary or HOST_SO)
        self._lib.shasta_mi355x_host_last_error.restype = C.c_char_p
        toc, data = meta_data
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        data = np.ascontiguousarray(data, dtype=np.uint8)
        self.read_count = len(toc) - 1
        self.delta, self.threads = int(delta), int(threads)
        self._keys = np.zeros(3 * max(1, self.read_count), dtype=np.uint64)          # 24 bytes per read
        rc = self._lib.shasta_mi355x_host_suppression_keys(toc.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p),
                                                            C.c_uint64(self.read_count), self._keys.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError(self._lib.shasta_mi355x_host_last_error().decode())

    def apply(self, candidates):
        """-> the candidates that stay: a view of a buffer this object keeps (valid until its next call; no 30 MB of fresh pages per call)."""
        import numpy as np
        candidates = np.ascontiguousarray(candidates)
        if getattr(self, "_out", None) is None or len(self._out) < len(candidates) or self._out.dtype != candidates.dtype:
            self._out = np.empty(len(candidates) + len(candidates) // 8 + 1, dtype=candidates.dtype)
        out = self._out[:len(candidates)]
        kept = C.c_uint64()
        rc = self._lib.shasta_mi355x_host_suppress_candidates_by_keys(
            self._keys.ctypes.data_as(C.c_void_p), C.c_uint64(self.read_count), candidates.ctypes.data_as(C.c_void_p), C.c_uint64(len(candidates)),
            out.ctypes.data_as(C.c_void_p), C.c_uint64(self.delta), C.c_uint64(self.threads), C.byref(kept))
        if rc != 0:
            raise RuntimeError(self._lib.shasta_mi355x_host_last_error().decode())
        return out[:int(kept.value)]
