/*
 * shasta_mi355x.h -- C ABI of libshasta_mi355x.so
 *
 * MI355X-native (gfx950, hand-written HIP) drop-in for the two hot functions of
 * the Shasta assembler's overlap detection (citations are /root/reference/ paths):
 *
 *   seam 1  LowHash0::LowHash0(...)                    src/LowHash0.hpp:32-52, src/LowHash0.cpp:23-257
 *           <- Assembler::findAlignmentCandidatesLowHash0   src/AssemblerLowHash.cpp:10-55
 *   seam 2  Assembler::computeAlignments(opts, threads)     src/Assembler.hpp:264-270, src/AssemblerAlign.cpp:208-304
 *           with alignMethod 4 (Align4::align, src/Align4.hpp:88-95, src/Align4.cpp:30)
 *
 * Plain C: pointers, sizes, PODs.  No C++/torch types cross this boundary.
 * Inputs are borrowed for the duration of a call (they are PROT_READ mmaps of
 * Shasta's Data/ files in the real caller).  Output buffers are owned by the
 * library and released with the matching *_free call.  Every function returns
 * 0 on success, non-zero on failure; shasta_mi355x_last_error() returns the
 * message for the calling thread (the C++ adapter rethrows std::runtime_error,
 * matching SHASTA_ASSERT / src/SHASTA_ASSERT.cpp:18-29).
 *
 * There is NO CPU fallback behind this ABI: if no gfx950 device is usable every
 * compute entry point fails with an error.
 */
#ifndef SHASTA_MI355X_H
#define SHASTA_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* Layout-compatible PODs                                                     */
/* ------------------------------------------------------------------------- */

/* shasta::OrientedReadPair, src/OrientedReadPair.hpp:18-86 (12 bytes, 3 pad). */
typedef struct shasta_oriented_read_pair {
    uint32_t readIds[2];     /* readIds[0] < readIds[1]                         */
    uint8_t  isSameStrand;   /* bool                                            */
    uint8_t  pad[3];         /* written as 0 by this library                    */
} shasta_oriented_read_pair;

/* shasta::AlignmentInfo, src/Alignment.hpp:86-447 (52 bytes). */
typedef struct shasta_alignment_info {
    struct {
        uint32_t markerCount;    /* markers in this oriented read                */
        uint32_t firstOrdinal;
        uint32_t lastOrdinal;
    } data[2];
    uint32_t markerCount;        /* aligned markers                              */
    int32_t  minOrdinalOffset;
    int32_t  maxOrdinalOffset;
    int32_t  averageOrdinalOffset;
    uint32_t maxSkip;
    uint32_t maxDrift;
    uint8_t  isInReadGraph;      /* bit 0; cleared                               */
    uint8_t  pad[3];
} shasta_alignment_info;

/* shasta::AlignmentData, src/Alignment.hpp:393-421 (64 bytes). */
typedef struct shasta_alignment_data {
    shasta_oriented_read_pair pair;
    shasta_alignment_info     info;
} shasta_alignment_data;

/* Arguments of LowHash0::LowHash0 that are numbers (src/LowHash0.hpp:32-43),
 * in the same order and with the same meaning. */
typedef struct shasta_lowhash0_params {
    uint64_t m;                          /* consecutive markers per feature      */
    double   hashFraction;
    uint64_t minHashIterationCount;      /* 0 => alignmentCandidatesPerRead rules */
    double   alignmentCandidatesPerRead;
    uint64_t log2MinHashBucketCount;     /* 0 => 5 + log2 estimate (:73-98)       */
    uint64_t minBucketSize;
    uint64_t maxBucketSize;
    uint64_t minFrequency;
} shasta_lowhash0_params;

/* What LowHash0 leaves behind besides the two memory mapped outputs: the
 * per-iteration console line (src/LowHash0.cpp:193-196) and the rows of
 * LowHashBucketHistogram.csv (:566-613). */
typedef struct shasta_lowhash0_result {
    uint64_t candidateCount;
    shasta_oriented_read_pair* candidates;  /* (readId0, readId1, strand) order  */
    uint32_t log2BucketCount;               /* the value actually used           */
    uint32_t iterationCount;                /* iterations actually run           */
    uint64_t* highFrequency;                /* [iterationCount]                  */
    uint64_t* total;                        /* [iterationCount]                  */
    uint64_t histogramRowCount;
    uint64_t* histogram;                    /* rows {iteration,bucketSize,bucketCount} */
    double   seconds;                       /* wall time of the call             */
    double   deviceSeconds;                 /* HIP-event time of the device part */
} shasta_lowhash0_result;

/* Align4::Options (src/Align4.hpp:106-121) plus the two outer switches of
 * computeAlignmentsThreadFunction (src/AssemblerAlign.cpp:316-331). */
typedef struct shasta_align4_options {
    uint64_t deltaX;
    uint64_t deltaY;
    uint64_t minEntryCountPerCell;
    uint64_t maxDistanceFromBoundary;
    uint64_t minAlignedMarkerCount;
    double   minAlignedFraction;
    uint64_t maxSkip;
    uint64_t maxDrift;
    uint64_t maxTrim;
    uint64_t maxBand;
    int64_t  matchScore;      /* carried for fidelity; Align4 itself always uses */
    int64_t  mismatchScore;   /* 6/-1/-1 (src/Align4.hpp:159-161: the members are */
    int64_t  gapScore;        /* never assigned from Options).                   */
    uint8_t  suppressContainments;
    uint8_t  pad[7];
} shasta_align4_options;

/* Align method 3 (the default of the shipped configurations): the numeric arguments of
 * Assembler::alignOrientedReads3 (src/AssemblerAlign3.cpp:22-33; defaults
 * src/AssemblerOptions.cpp:419-449), k (the marker length: the down-sampling hash
 * KmerInfo::hash is MurmurHash2 of kmerId + its reverse complement,
 * src/AssemblerKmers.cpp:182-186) and the outer filters of computeAlignmentsThreadFunction
 * (src/AssemblerAlign.cpp:439-472). */
typedef struct shasta_align3_options {
    int64_t  matchScore;
    int64_t  mismatchScore;
    int64_t  gapScore;
    double   downsamplingFactor;     /* fraction of the markers kept in step 1      */
    int64_t  bandExtend;             /* step-2 band = step-1 offset range +- this   */
    int64_t  maxBand;                /* wider step-2 bands give an empty alignment  */
    uint64_t k;
    uint64_t minAlignedMarkerCount;
    double   minAlignedFraction;
    uint64_t maxSkip;
    uint64_t maxDrift;
    uint64_t maxTrim;
    uint8_t  suppressContainments;
    uint8_t  pad[7];
} shasta_align3_options;

/* Per-candidate status codes. */
enum {
    SHASTA_ALIGN_STORED        = 0,  /* passed every filter; one AlignmentData row  */
    SHASTA_ALIGN_REJECTED      = 1,  /* aligned but failed a filter (:439-472)      */
    SHASTA_ALIGN_EMPTY         = 2,  /* Align4 found no acceptable component        */
    SHASTA_ALIGN_SKIPPED       = 3,  /* resource limit: the reference's "skip+log"  */
                                     /* lane (src/AssemblerAlign.cpp:419-435)       */
    SHASTA_ALIGN_TIE_FLAG      = 0x80 /* or-ed in: two components tied on markerCount */
                                     /* (the reference takes the first in its union- */
                                     /* find order, src/Align4.cpp:792-872) and the  */
                                     /* tie could not be resolved the reference's way */
                                     /* (normally it is: DESIGN.md section 2)         */
};

typedef struct shasta_align4_result {
    uint64_t alignmentCount;
    shasta_alignment_data* alignmentData;    /* [alignmentCount], candidate order   */
    uint64_t* compressedToc;                 /* [alignmentCount+1]                  */
    uint8_t*  compressedData;                /* shasta::compress bytes              */
    uint8_t*  status;                        /* [candidateCount]                    */
    /* Diagnostics for parity tests: the alignment Align4 returned for EVERY
     * candidate (before the outer filters).  ordinals are (x,y) pairs. */
    uint64_t* ordinalsToc;                   /* [candidateCount+1], in pairs        */
    uint32_t* ordinals;                      /* 2 * ordinalsToc[candidateCount]     */
    uint64_t  dpCellCount;                   /* sum of nx * bandWidth over DPs run  */
    uint64_t  kmerIdBytes;                   /* sum of 4*(nx+ny) over candidates    */
    uint64_t  alignedBytes;                  /* sum of 8*markerCount over the stored alignments (8a) */
    double    seconds;
    double    deviceSeconds;
    void*     owner;        /* NULL: the arrays are released by shasta_mi355x_align4_free; otherwise they
                               belong to that context (align4_run_borrowed) and free only clears the struct */
} shasta_align4_result;

/* ------------------------------------------------------------------------- */
/* Library / device                                                           */
/* ------------------------------------------------------------------------- */

const char* shasta_mi355x_last_error(void);
const char* shasta_mi355x_version(void);
/* Number of usable gfx950 devices (0 if none; never throws). */
int shasta_mi355x_device_count(void);
/* What in the process's environment costs the library speed, as bits (0: nothing): bit 0 = GPU_MAX_HW_QUEUES is not set -- the HIP
 * runtime then deals the aligner's streams to four hardware queues and a call of the second seam is about 15 % slower (set it to 8 before
 * the process's first HIP call).  The library says the same once on stderr when a context or a group is created. */
int shasta_mi355x_environment_warnings(void);

/* ------------------------------------------------------------------------- */
/* One-shot, host-pointer seams (what the C++ adapter calls)                  */
/* ------------------------------------------------------------------------- */

/* Seam 1.  markersToc has 2*readCount+1 entries (Markers.toc), markersData is
 * Markers.data: packed 7-byte CompressedMarker {u32 kmerId, u24 position}
 * (src/Marker.hpp:56-70).  readFlags is ReadFlags (1 byte per read, bit 0 =
 * isPalindromic, src/ReadFlags.hpp:10-30).  readLowHashStatistics receives
 * readCount*3 u64 {sparse, good, crowded}, zeroed by the callee
 * (src/LowHash0.cpp:123-125). */
int shasta_mi355x_lowhash0(
    uint64_t readCount,
    const uint64_t* markersToc,
    const void* markersData,
    const uint8_t* readFlags,
    const shasta_lowhash0_params* params,
    uint64_t* readLowHashStatistics,
    shasta_lowhash0_result* result);
void shasta_mi355x_lowhash0_free(shasta_lowhash0_result* result);

/* Seam 2 (method 4).  Aligns candidates[i] = (readId0 strand 0, readId1 strand
 * isSameStrand?0:1), applies the filters of src/AssemblerAlign.cpp:439-472 and
 * returns AlignmentData + CompressedAlignments in candidate order (the
 * reference's order for threadCount==1). */
int shasta_mi355x_align4_batch(
    uint64_t readCount,
    const uint64_t* markersToc,
    const void* markersData,
    uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates,
    const shasta_align4_options* options,
    int wantOrdinals,
    shasta_align4_result* result);
void shasta_mi355x_align4_free(shasta_align4_result* result);

/* ------------------------------------------------------------------------- */
/* Device-resident context (bench, multi-GPU driver, staged tests)            */
/* ------------------------------------------------------------------------- */

typedef struct shasta_mi355x_ctx shasta_mi355x_ctx;

/* Creates a context on HIP device `device` with its own non-blocking stream. */
shasta_mi355x_ctx* shasta_mi355x_create(int device);
void shasta_mi355x_destroy(shasta_mi355x_ctx*);

/* Uploads Markers.{toc,data} and runs the marker-strip kernel
 * (LowHash0::createKmerIds, src/LowHash0.cpp:261-308) so that the dense
 * uint32 kmerIds[M] + toc live in HBM.  readFlags may be NULL (all zero). */
int shasta_mi355x_set_markers(
    shasta_mi355x_ctx*, uint64_t readCount,
    const uint64_t* markersToc, const void* markersData, const uint8_t* readFlags);

/* Adopts kmer ids that are already dense on the HOST (uint32 per marker); used
 * by synthetic benchmarks that never materialise 7-byte markers. */
int shasta_mi355x_set_kmer_ids(
    shasta_mi355x_ctx*, uint64_t readCount,
    const uint64_t* markersToc, const uint32_t* kmerIds, const uint8_t* readFlags);

/* Same, with the dense kmer ids already in HBM on this context's device (e.g. the output of
 * an all-gather of the ranks' shards). */
int shasta_mi355x_set_kmer_ids_device(
    shasta_mi355x_ctx*, uint64_t readCount,
    const uint64_t* markersToc, const void* kmerIdsDevice, const uint8_t* readFlags);

/* LowHash0 on the resident markers.  Same outputs as the one-shot seam. */
int shasta_mi355x_lowhash0_run(
    shasta_mi355x_ctx*, const shasta_lowhash0_params* params,
    uint64_t* readLowHashStatistics, shasta_lowhash0_result* result);

/* Align4 batch on the resident markers. */
int shasta_mi355x_align4_run(
    shasta_mi355x_ctx*, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates,
    const shasta_align4_options* options, int wantOrdinals,
    shasta_align4_result* result);

/* -------------------------------------------------------------------------
 * LowHash0 in stages, for one job sharded over several GPUs.
 * One process per GPU; every rank holds all markers (set_markers / set_kmer_ids*), hashes the
 * reads of its own range, owns the bucket ids [rank, rank+1) * 2^log2 / worldSize and the pair
 * keys whose readId0 lies in its read range.  Between the stages the CALLER moves data between
 * ranks (RCCL all-to-all over xGMI in shasta_amd/distributed.py); this library never talks to
 * another process.  Pointers named *Device are device memory of this context's GPU, valid until
 * the next stage call.  readBoundaries has worldSize+1 read ids, [0 .. readCount].
 *   lh_begin   -> log2BucketCount
 *   per iteration (same iteration control as src/LowHash0.cpp:136-157, on all-reduced counts):
 *     lh_hash     K1+K2: records {bucketId u32} / {hashHigh:32 | orientedReadId:32 u64}, sorted by
 *                 bucket id; sendOffsets[r..r+1] = the records rank r owns
 *       -- all-to-all of records --
 *     lh_buckets  K3 + K4 on the n received records: statistics (partial sums kept on the device),
 *                 bucket-size histogram bins [0,2048) + list of larger sizes, bucketsUsed, and this
 *                 iteration's pair keys {u64: readId0 | readId1 | strand bit}, sorted; sendOffsets by
 *                 owner of readId0
 *       -- all-to-all of pair keys (8 bytes each) --
 *     lh_merge    appends the n received keys to this rank's keys of all iterations.  With evaluateNow
 *                 (needed after every iteration only when minHashIterationCount = 0) it also evaluates
 *                 them: this rank's share of the latest iteration's "high frequency" and "total"
 *                 counters (all-reduce them); otherwise both come back 0
 *   or, with a fixed number of iterations (minHashIterationCount > 0), ALL iterations in one pass -- one call of each
 *   stage and two exchanges per JOB instead of per iteration (fewer, larger collectives; the markers are read once):
 *     lh_hash_all     records of every iteration {u64: owner << 56 | iteration << 32 | bucketId} / {u64 as above},
 *                     each owner's records contiguous; sendOffsets[r..r+1] = the records rank r owns
 *       -- all-to-all of records --
 *     lh_buckets_all  on the n received records: statistics, bucketsUsed[iteration], sizeHistogram[iteration][2048],
 *                     the larger sizes as iteration << 32 | size, and the pair keys of all iterations with their
 *                     iteration tags {u32}, sorted by key; sendOffsets by owner of readId0
 *       -- all-to-all of pair keys and tags --
 *     lh_merge_all    takes the n received keys and tags
 *   lh_finish  K5 + K6: evaluates the keys of all iterations at once; this rank's candidates (sorted;
 *              concatenating the ranks in order gives the reference's order) -- free with
 *              shasta_mi355x_free --, its partial readLowHashStatistics[readCount*3] and its share of
 *              the per-iteration "high frequency" / "total" counters (all-reduce all three).
 *   lh_finish_on_device  the same with the candidates left in device memory (*candidatesDevice, 12-byte records; valid until
 *              this context's next lh_begin / lowhash0_run): the caller's all-gather of the ranks' lists reads them where
 *              they are instead of after a copy to the host and back.
 * ------------------------------------------------------------------------- */
#define SHASTA_MI355X_SIZE_HISTOGRAM_BINS 2048
int shasta_mi355x_lh_begin(shasta_mi355x_ctx*, const shasta_lowhash0_params* params, int rank, int worldSize,
    const uint64_t* readBoundaries, uint32_t* log2BucketCount);
/* After lh_begin: *fits = 1 if this rank can take the job's iterations in one pass (lh_hash_all / lh_buckets_all / lh_merge_all):
 * a fixed number of iterations (1 .. 4096), at most 256 ranks, and the low-hash records of ALL iterations within the 2^32 positions of
 * one sort with a factor of two to spare.  The caller reduces the answers of all ranks (minimum) and takes the one-pass form only
 * if every rank said 1; otherwise lh_hash / lh_buckets / lh_merge per iteration, which need one iteration's records to fit. */
int shasta_mi355x_lh_one_pass_fits(shasta_mi355x_ctx*, int* fits);
int shasta_mi355x_lh_hash(shasta_mi355x_ctx*, uint64_t iteration, uint64_t* sendOffsets,
    const void** keysDevice, const void** valsDevice);
int shasta_mi355x_lh_buckets(shasta_mi355x_ctx*, const void* keysDevice, const void* valsDevice, uint64_t n,
    uint64_t* sendOffsets, const void** pairKeysDevice, uint64_t* bucketsUsed,
    uint64_t* sizeHistogram, uint32_t* overflowSizes, uint64_t overflowCapacity, uint64_t* overflowCount);
int shasta_mi355x_lh_merge(shasta_mi355x_ctx*, const void* pairKeysDevice, uint64_t n, int evaluateNow,
    uint64_t* highFrequency, uint64_t* total);
int shasta_mi355x_lh_hash_all(shasta_mi355x_ctx*, uint64_t* sendOffsets, const void** keysDevice, const void** valsDevice);
int shasta_mi355x_lh_buckets_all(shasta_mi355x_ctx*, const void* keysDevice, const void* valsDevice, uint64_t n,
    uint64_t* sendOffsets, const void** pairKeysDevice, const void** pairTagsDevice, uint64_t iterationCapacity, uint64_t* bucketsUsed,
    uint64_t* sizeHistogram, uint64_t* overflowSizes, uint64_t overflowCapacity, uint64_t* overflowCount);
int shasta_mi355x_lh_merge_all(shasta_mi355x_ctx*, const void* pairKeysDevice, const void* pairTagsDevice, uint64_t n);
int shasta_mi355x_lh_finish(shasta_mi355x_ctx*, uint64_t* readLowHashStatistics,
    shasta_oriented_read_pair** candidates, uint64_t* candidateCount,
    uint64_t* highFrequencyPerIteration, uint64_t* totalPerIteration, uint64_t iterationCapacity, uint64_t* iterationCount);
int shasta_mi355x_lh_finish_on_device(shasta_mi355x_ctx*, uint64_t* readLowHashStatistics,
    const shasta_oriented_read_pair** candidatesDevice, uint64_t* candidateCount,
    uint64_t* highFrequencyPerIteration, uint64_t* totalPerIteration, uint64_t iterationCapacity, uint64_t* iterationCount);
void shasta_mi355x_free(void*);
/* Synchronous copy on the context's stream; kind 0 host->device, 1 device->host, 2 device->device. */
int shasta_mi355x_memcpy(shasta_mi355x_ctx*, void* dst, const void* src, uint64_t bytes, int kind);

/* Same as align4_run, but the result arrays belong to the context: they stay valid until the next
 * align4_run_borrowed on this context (or its destruction) and shasta_mi355x_align4_free only
 * clears the struct.  For callers that consume the result at once (the C++ adapter copies it into
 * the memory mapped vectors): saves allocating, faulting in and unmapping the result every call. */
int shasta_mi355x_align4_run_borrowed(
    shasta_mi355x_ctx*, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates,
    const shasta_align4_options* options, int wantOrdinals,
    shasta_align4_result* result);

/* Align method 3 (Assembler::alignOrientedReads3, src/AssemblerAlign3.cpp:22-314, called from
 * computeAlignmentsThreadFunction, src/AssemblerAlign.cpp:404-409) for every candidate, then the
 * same filters and outputs as the method-4 seam; results are released with
 * shasta_mi355x_align4_free.  Status EMPTY = empty alignment (a read without down-sampled markers,
 * nothing aligned in step 1, or a band wider than maxBand); SKIPPED = a pair whose down-sampled
 * matrix has more than 65536 diagonals (down-sampled markers of the two reads + 1: reads of several
 * megabases at the default factor), which this version does not align.  Limits: any scores whose magnitudes
 * add up to less than 2^20 (6/-1/-1, every shipped configuration, runs the kernels with the scores as
 * immediates), maxBand <= 65535 (a band of more than 1024 diagonals runs in the wide DP), k <= 16.
 * `borrowed` != 0: result arrays belong to the context, as align4_run_borrowed. */
int shasta_mi355x_align3_run(
    shasta_mi355x_ctx*, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates,
    const shasta_align3_options* options, int wantOrdinals, int borrowed,
    shasta_align4_result* result);
/* One-shot form on host markers (what the C++ adapter calls for alignMethod 3). */
int shasta_mi355x_align3_batch(
    uint64_t readCount,
    const uint64_t* markersToc,
    const void* markersData,
    uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates,
    const shasta_align3_options* options,
    int wantOrdinals,
    shasta_align4_result* result);

/* -------------------------------------------------------------------------
 * Seam 0, the producer of the path's input: marker finding,
 * MarkerFinder::MarkerFinder (src/MarkerFinder.hpp:25-31, src/MarkerFinder.cpp:16-127) as called by
 * Assembler::findMarkers (src/AssemblerMarkers.cpp:11-24).
 *   reads       Shasta's LongBaseSequences (Data/Reads-Bases.{toc,data}, Data/Reads-BaseCount):
 *               readsToc[readCount+1] are word offsets into readsData; a read of n bases has
 *               2*ceil(n/64) words, per 64 bases the low bits then the high bits of the bases,
 *               base 0 in the most significant bit (src/LongBaseSequence.hpp:33-41)
 *   kmerTable   4^k entries of kmerInfoStride bytes; the byte at isMarkerOffset is KmerInfo::isMarker
 *               (Data/Kmers: stride 24, offset 12; src/Kmer.hpp:22-39)
 * Output: markersToc[2*readCount+1] and, if wantPacked, Markers.data (7-byte CompressedMarker
 * records), both released by shasta_mi355x_find_markers_free.  ctx may be NULL (a temporary context
 * is used); with a context the markers stay resident exactly as after shasta_mi355x_set_markers, so
 * that LowHash0 and the aligners run without uploading 7 bytes per marker (readFlags as there).
 * ------------------------------------------------------------------------- */
typedef struct shasta_markers_result {
    uint64_t  markerCount;         /* both strands                                */
    uint64_t* markersToc;          /* [2*readCount+1]                             */
    uint8_t*  markersData;         /* 7*markerCount bytes, or NULL                */
    double    seconds;
    double    deviceSeconds;
} shasta_markers_result;
int shasta_mi355x_find_markers(
    shasta_mi355x_ctx* ctx, uint64_t readCount,
    const uint64_t* readsToc, const uint64_t* readsData, const uint64_t* baseCounts,
    uint64_t k, const void* kmerTable, uint64_t kmerInfoStride, uint64_t isMarkerOffset,
    const uint8_t* readFlags, int wantPacked, shasta_markers_result* result);
void shasta_mi355x_find_markers_free(shasta_markers_result* result);

/* Per-kernel timing table of a context, for bench reports (no counterpart in the reference): every kernel
 * launch of the stages is bracketed by two HIP events on the stream it is issued on.  One row per kernel
 * name, accumulated over all calls on the context since the last reset: seconds = sum of the event
 * durations of its launches; algorithmicBytes = sum over launches of SURVEY 8(d)'s per-unit figure x the
 * units of the launch (0 where no such figure is defined); work = a kernel-specific unit count (windows
 * hashed, DP cells nx x bandWidth, candidates, aligned markers ...).
 * shasta_mi355x_kernel_table writes min(*count, capacity) rows and sets *count to the number of rows. */
typedef struct shasta_mi355x_kernel_stat {
    char     name[64];
    double   seconds;
    uint64_t launches;
    uint64_t algorithmicBytes;
    uint64_t work;
} shasta_mi355x_kernel_stat;
int shasta_mi355x_kernel_table(shasta_mi355x_ctx*, shasta_mi355x_kernel_stat* rows, uint64_t capacity, uint64_t* count);
int shasta_mi355x_kernel_table_reset(shasta_mi355x_ctx*);

/* ------------------------------------------------------------------------- */
/* Unit seams used by the parity tests                                        */
/* ------------------------------------------------------------------------- */

/* Profiling aid, no counterpart in the reference: moves exactly `bytes` through HBM with the
 * access pattern of the window-hash kernel (mode 0: dword reads) or of the DP trace (mode 1:
 * 8-byte record stores), so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be calibrated on a known
 * byte count (MI355X_MICROARCH.md, HBM section). */
int shasta_mi355x_calibrate(uint64_t bytes, int mode);

/* MurmurHash64A (src/MurmurHash2.cpp:96-140) of every window of m kmer ids,
 * seed = 37*iteration: out[i] for i in [0, n-m+1).  Device computes; host
 * pointers in and out. */
int shasta_mi355x_hash_windows(
    const uint32_t* kmerIds, uint64_t n, uint64_t m, uint64_t iteration, uint64_t* out);

/* Banded overlap DP + traceback on one pair (src/Align4.cpp:993-1088).  Writes
 * the diagonal-and-equal steps as (x,y) pairs; returns their number in *count
 * (capacity is in pairs). */
int shasta_mi355x_banded_dp(
    const uint32_t* kmerIds0, uint32_t nx,
    const uint32_t* kmerIds1, uint32_t ny,
    int32_t bandMin, int32_t bandMax,
    uint32_t* ordinals, uint64_t capacity, uint64_t* count, int32_t* score);

/* The same DP on many tasks in one call: they are sorted by band class and length and bundled several to a
 * wavefront exactly as the tasks of an Align4 batch are.  Task t aligns kmerIds[begin0[t] .. +nx[t]) with
 * kmerIds[begin1[t] .. +ny[t]) inside the band [bandMin[t], bandMax[t]] (width <= 65536, meeting the matrix; more than 1024 diagonals: the wide DP).
 * counts[t] aligned pairs and scores[t] per task; the pairs of all tasks concatenated in ordinals
 * (capacity in pairs; ordinals may be NULL).  seconds (NULL or 9 entries): HIP-event time of the forward
 * launch of each of the eight band classes (widths <= 32, 48, 64, 80, 128, 256, 512, 1024) and of the traceback;
 * cells (NULL or 8 entries): DP cells (nx x band width) per class.  A unit seam for parity tests and for
 * timing one kernel version against another (scripts/dp_microbench.py), like shasta_mi355x_banded_dp. */
int shasta_mi355x_banded_dp_many(
    const uint32_t* kmerIds, uint64_t kmerCount, uint64_t taskCount,
    const uint64_t* begin0, const uint32_t* nx, const uint64_t* begin1, const uint32_t* ny,
    const int32_t* bandMin, const int32_t* bandMax,
    uint64_t* counts, int32_t* scores, uint32_t* ordinals, uint64_t capacity,
    double* seconds, uint64_t* cells);

/* -------------------------------------------------------------------------
 * The tables either side of the aligner, device = HIP device index.
 * shasta_mi355x_pair_table: per oriented read, the indices of the pairs it takes part in, sorted by (other
 * oriented read, index) -- Assembler::computeAlignmentTable (src/AssemblerAlign.cpp:509-571; pairs = the
 * AlignmentData rows, strideBytes = 64) and AlignmentCandidates::computeCandidateTable
 * (src/AssemblerAlignmentCandidates.cpp:388-447; pairs = the candidates, strideBytes = 12).  Every pair appears
 * under its two oriented reads and under their reverse complements.  toc: 2 readCount + 1 offsets; values:
 * 4 pairCount indices.  pairCount < 2^32 (the indices are 32-bit); a list of more than 2^29 pairs is sorted range by range
 * of oriented reads (one radix sort takes 2^31 entries), the pairs passing through the device in slabs.
 * shasta_mi355x_read_graph_keep: createReadGraph's selection (src/AssemblerReadGraph.cpp:55-95): keep[i] = 1 if
 * alignment i is among the maxAlignmentCount alignments with the largest (markerCount, index) of either of
 * its reads, else 0. */
int shasta_mi355x_pair_table(int device, const void* pairs, uint64_t strideBytes, uint64_t pairCount, uint64_t readCount,
    uint64_t* toc, uint32_t* values);
/* The last step of Assembler::computeAlignments (src/AssemblerAlign.cpp:296 -> computeAlignmentTable, :509-571) for the
 * alignments the context's last shasta_mi355x_align4_run_borrowed / _align3_run_borrowed call stored: the same table as
 * shasta_mi355x_pair_table gives for those AlignmentData rows (toc: 2 readCount + 1 offsets, values: 4 alignmentCount
 * indices), built where the aligner left off -- the context's stream and buffers, nothing allocated from the second call on.
 * *toc and *values point into page-locked arrays of the context, valid until its next aligner or table call. */
int shasta_mi355x_alignment_table(shasta_mi355x_ctx*, const uint64_t** toc, const uint32_t** values, uint64_t* valueCount);
int shasta_mi355x_read_graph_keep(int device, const shasta_alignment_data* alignmentData, uint64_t alignmentCount, uint64_t readCount,
    uint32_t maxAlignmentCount, uint8_t* keep);

/* -------------------------------------------------------------------------
 * Palindromic-read flagging: the device half of
 * Assembler::flagPalindromicReads (src/AssemblerAlign.cpp:652-770, parameters
 * src/AssemblerOptions.cpp:255-288).  For every read of the context's markers:
 * the number of marker pairs (ordinal i on strand 0, ordinal j on strand 1) with
 * equal kmer ids and |i - j| < deltaThreshold.  Every marker of the reference's
 * method-0 self-alignment that it counts as near-diagonal (:738-749) is such a
 * pair, so a read with double(bound[r]) / double(markerCount) below
 * nearDiagonalFractionThreshold cannot be flagged (:750-752); the others are decided
 * by the host layer with the reference's own sequential graph search
 * (shasta_amd/host/PalindromicReads.cpp).  bound: host pointer, readCount entries.
 * deltaThreshold must be in [1, 4096].
 * ------------------------------------------------------------------------- */
int shasta_mi355x_palindromic_screen(shasta_mi355x_ctx*, uint64_t deltaThreshold, uint32_t* bound);

/* -------------------------------------------------------------------------
 * Several GPUs of one node behind ONE blocking call from one process -- the shape of the reference's seams
 * (src/AssemblerLowHash.cpp:36-52, src/AssemblerAlign.cpp:208-304 are single C++ calls that fan out over threads
 * inside; SURVEY 8(b) proposed `int nGpus` on both entry points).  devices: deviceCount HIP device ids, or NULL
 * for 0 .. deviceCount-1 (a device may be listed more than once: the sharded path then runs on fewer GPUs, which
 * is how it is tested on a one-GPU box).  Results are identical to the one-GPU entry points for any device count.
 * LowHash0: reads sharded by contiguous id ranges balanced by marker count, bucket ids and readId0 ranges owned by
 * device, the two exchanges of an iteration as device-to-device copies over xGMI (hipMemcpyPeerAsync, each device
 * pulls its segments), reductions on the host after the last iteration.  Aligners: contiguous candidate ranges
 * balanced by the markers they touch, results concatenated in candidate order.
 *   *_multi           one-shot forms of shasta_mi355x_lowhash0 / _align4_batch / _align3_batch (host pointers in)
 *   shasta_mi355x_group   the markers stay resident on every device of the group between calls
 * ------------------------------------------------------------------------- */
typedef struct shasta_mi355x_group shasta_mi355x_group;
shasta_mi355x_group* shasta_mi355x_group_create(int deviceCount, const int* devices);    /* NULL on error */
void shasta_mi355x_group_destroy(shasta_mi355x_group*);
int shasta_mi355x_group_set_markers(shasta_mi355x_group*, uint64_t readCount,
    const uint64_t* markersToc, const void* markersData, const uint8_t* readFlags);
int shasta_mi355x_group_set_kmer_ids(shasta_mi355x_group*, uint64_t readCount,
    const uint64_t* markersToc, const uint32_t* kmerIds, const uint8_t* readFlags);
int shasta_mi355x_group_lowhash0_run(shasta_mi355x_group*, const shasta_lowhash0_params*,
    uint64_t* readLowHashStatistics, shasta_lowhash0_result*);
int shasta_mi355x_group_align4_run(shasta_mi355x_group*, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates, const shasta_align4_options*, int wantOrdinals, shasta_align4_result*);
int shasta_mi355x_group_align3_run(shasta_mi355x_group*, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates, const shasta_align3_options*, int wantOrdinals, shasta_align4_result*);
/* The same with the result arrays owned by the group (result->owner != NULL: shasta_mi355x_align4_free only clears the struct):
 * valid until the group's next aligner call or its destruction.  A caller that copies the result into its own containers at
 * once -- Shasta's Assembler::computeAlignments appends to memory-mapped vectors, src/AssemblerAlign.cpp:262-283 -- saves the
 * allocation and first touch of half a gigabyte per call (a third of the call at 100 k reads). */
int shasta_mi355x_group_align4_run_borrowed(shasta_mi355x_group*, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates, const shasta_align4_options*, int wantOrdinals, shasta_align4_result*);
int shasta_mi355x_group_align3_run_borrowed(shasta_mi355x_group*, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates, const shasta_align3_options*, int wantOrdinals, shasta_align4_result*);
int shasta_mi355x_lowhash0_multi(uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    const uint8_t* readFlags, const shasta_lowhash0_params*, int deviceCount, const int* devices,
    uint64_t* readLowHashStatistics, shasta_lowhash0_result*);
int shasta_mi355x_align4_batch_multi(uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align4_options*, int wantOrdinals, int deviceCount, const int* devices, shasta_align4_result*);
int shasta_mi355x_align3_batch_multi(uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align3_options*, int wantOrdinals, int deviceCount, const int* devices, shasta_align4_result*);

#ifdef __cplusplus
}
#endif
#endif
