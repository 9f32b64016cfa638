// The device-resident state behind shasta_mi355x_ctx: Shasta's Markers laid out
// for HBM (dense uint32 kmerIds[M] + uint64 toc[2R+1] + 1-byte ReadFlags[R]),
// one HIP stream, and the grow-only workspaces of the two stages.
#pragma once

#include "common.hpp"
#include "primitives.hpp"
#include "../../include/shasta_mi355x.h"

#include <memory>
#include <mutex>
#include <vector>

namespace shasta_mi355x {

constexpr int HASH_TILE = 252;        // markers per hash-kernel tile: 63 lanes x 4 windows (the 64th lane holds the halo)

struct Context {
    int device = 0;
    hipStream_t stream = nullptr;

    // Markers in HBM.
    uint64_t readCount = 0;
    uint64_t markerCount = 0;                // both strands
    std::vector<uint64_t> hostToc;           // 2R+1
    DeviceBuffer<uint64_t> toc;              // 2R+1
    DeviceBuffer<uint32_t> kmerIds;          // M
    DeviceBuffer<uint8_t> readFlags;         // R
    DeviceBuffer<uint4> tileDesc;            // ceil(M/HASH_TILE)+1: {first oriented read, its palindromic flag, its end (u64)} per hash tile
    int matchShift = 13;                     // two random markers of the read set are equal about once in 2^(matchShift + 1) .. 2^(matchShift + 2) pairs (setMarkers' sample; never below 13)

    // The aligner runs its batches on several host workers (ALIGN_MAX_WORKERS at most), each with its own stream, sort
    // workspace, side stream (wide-band DP classes) and grow-only batch scratch: worker 0 uses `stream` / `sortWs`.
    static constexpr int ALIGN_MAX_WORKERS = 8;
    RadixSortWorkspace sortWs, workerSortWs[ALIGN_MAX_WORKERS];
    hipStream_t workerStream[ALIGN_MAX_WORKERS] = {};
    hipStream_t wideStream[ALIGN_MAX_WORKERS] = {};
    std::shared_ptr<void> alignScratch[ALIGN_MAX_WORKERS];
    SharedCapacities alignCapacities;        // high-water marks the workers' scratch buffers share (common.hpp)
    std::shared_ptr<void> alignStore;        // results of borrowed Align4 calls (valid until the next call)
    std::shared_ptr<void> lowhashJob;        // LowHash0 job in progress
    std::shared_ptr<void> lowhashBuffers;    // the last finished job, kept for its device allocations (the next job adopts them)
    uint64_t lowhashRecordsHint = 0, lowhashPairsHint = 0;   // capacities the last jobs needed (first guesses of the next)
    std::shared_ptr<void> tableStore;        // alignmentTableOfLastCall: its device buffers and the page-locked arrays the caller reads
    std::shared_ptr<void> downsampled;       // align method 3: the markers its step 1 keeps (dropped by setMarkers)
    // Kernels that need more dynamic LDS than the default get the attribute once per context, i.e. on this context's device
    // (hipFuncSetAttribute acts on the current device; the aligner's workers may get there at the same time).
    std::once_flag cellsLdsAttribute[4], cellsDumpLdsAttribute, wideDpLdsAttribute, palindromicLdsAttribute;      // per context = per device (hipFuncSetAttribute is per device)
    KernelTimers timers;                     // per-kernel HIP-event times since the last reset (shasta_mi355x_kernel_table)
    // The aligner's own events (a call's begin / end / join, two per worker): made once and kept -- fifteen hipEventCreate at the head
    // of every call and fifteen hipEventDestroy at its end were 4 + 5-10 ms of a 130 ms call on the host's clock (round 5,
    // SHASTA_MI355X_LOG_HOST=1).  Index: 0-2 the call's, 3 + 2 k and 4 + 2 k worker k's.
    std::vector<hipEvent_t> alignEvents;
    hipEvent_t alignEvent(size_t index)
    {
        while(alignEvents.size() <= index) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); alignEvents.push_back(e); }
        return alignEvents[index];
    }

    explicit Context(int device);
    ~Context();
    void setMarkers(uint64_t readCount, const uint64_t* toc, const void* data7,
        const uint32_t* denseKmerIds, const uint8_t* flags, bool denseOnDevice = false);
};

// Several devices of one node behind one call (multi.hip): one context per device, one host thread per device for the
// duration of a call, device-to-device copies over xGMI between the stages of LowHash0.
struct Group {
    std::vector<std::unique_ptr<Context>> contexts;
    Group(int deviceCount, const int* devices);          // devices == nullptr: 0 .. deviceCount - 1; a device may be listed more than once
    void setMarkers(uint64_t readCount, const uint64_t* toc, const void* data7, const uint32_t* denseKmerIds, const uint8_t* flags);
    void lowhash0Run(const shasta_lowhash0_params&, uint64_t* readLowHashStatistics, shasta_lowhash0_result&);
    // borrowed: the result's arrays belong to the group (result.owner = the group) and are reused by its next aligner call.
    void alignRun(uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
        const shasta_align4_options* options4, const shasta_align3_options* options3, bool wantOrdinals, shasta_align4_result&, bool borrowed = false);
    ~Group();
    // The exchanges' second transport (multi.hip): RCCL, one communicator per device of the group (ncclCommInitAll), grouped
    // ncclSend / ncclRecv over xGMI.  Created at the first call that asks for it; empty: the peer-copy transport.
    std::vector<void*> rcclCommunicators;
    bool rcclTried = false;
    std::atomic<bool> rcclAborted{false};      // a rank's RCCL call failed and the communicators were aborted (multi.hip)
    // The concatenated result of a borrowed call over several devices.
    std::vector<shasta_alignment_data> storeRows;
    std::vector<uint64_t> storeToc, storeOrdinalsToc;
    std::vector<uint8_t> storeBytes, storeStatus;
    std::vector<uint32_t> storeOrdinals;
};

// One timed launch (or group of launches that form one step): events on `stream` around the statement(s).
#define SHASTA_TIMED(ctx, name, stream, bytes, work, ...) do { const KernelTimers::Span span_ = (ctx).timers.begin(name, stream); \
    __VA_ARGS__; (void)(ctx).timers.end(span_, bytes, work); } while(0)

// Stage entry points (lowhash0.hip, align4.hip).
void lowhash0Run(Context&, const shasta_lowhash0_params&, uint64_t* readLowHashStatistics, shasta_lowhash0_result&);
void lowhash0Free(shasta_lowhash0_result&);
// Staged form of the same job (one GPU or one rank of several); device pointers in and out.
constexpr int LOWHASH0_SIZE_HISTOGRAM_BINS = 2048;
void lowhash0Begin(Context&, const shasta_lowhash0_params&, int rank, int world, const uint64_t* readBoundaries, uint32_t* log2BucketCount);
void lowhash0Hash(Context&, uint64_t iteration, uint64_t* sendOffsets, const uint32_t** keys, const uint64_t** vals);
void lowhash0Buckets(Context&, const uint32_t* keys, const uint64_t* vals, uint64_t n, uint64_t* sendOffsets,
    const uint64_t** pairKeys, uint64_t* bucketsUsed, uint64_t* sizeHistogram, std::vector<uint32_t>& overflow);
void lowhash0Merge(Context&, const uint64_t* pairKeys, uint64_t n, bool evaluateNow, uint64_t* highFrequency, uint64_t* total);
// The same job with all iterations in one pass (fixed minHashIterationCount): one call of each per job.
bool lowhash0OnePassFits(Context&);           // after lowhash0Begin: may this rank take the one-pass form?  (all ranks must agree)
void lowhash0HashAll(Context&, uint64_t* sendOffsets, const uint64_t** keys, const uint64_t** vals);
void lowhash0BucketsAll(Context&, const uint64_t* keys, const uint64_t* vals, uint64_t n, uint64_t* sendOffsets,
    const uint64_t** pairKeys, const uint32_t** pairTags, uint64_t* bucketsUsed, uint64_t* sizeHistogram, std::vector<uint64_t>& overflow);
void lowhash0MergeAll(Context&, const uint64_t* pairKeys, const uint32_t* pairTags, uint64_t n);
uint64_t lowhash0JobIterations(Context&);
uint64_t lowhash0JobPlannedIterations(Context&);          // minHashIterationCount of the job in progress
void lowhash0Finish(Context&, uint64_t* readLowHashStatistics, std::vector<shasta_oriented_read_pair>& candidates,
    std::vector<uint64_t>& highFrequencyPerIteration, std::vector<uint64_t>& totalPerIteration);
void lowhash0Finish(Context&, uint64_t* readLowHashStatistics, std::vector<shasta_oriented_read_pair>* candidatesOrNull,
    std::vector<uint64_t>& highPerIteration, std::vector<uint64_t>& totalPerIteration,
    const shasta_oriented_read_pair** deviceCandidates, uint64_t* deviceCandidateCount);
void align4Run(Context&, uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align4_options&, bool wantOrdinals, shasta_align4_result&, bool borrowed = false);
void align3Run(Context&, uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align3_options&, bool wantOrdinals, shasta_align4_result&, bool borrowed = false);
void align4Free(shasta_align4_result&);
// markers.hip: MarkerFinder on the device; the context holds the markers afterwards.
void findMarkers(Context&, uint64_t readCount, const uint64_t* readsToc, const uint64_t* readsData, const uint64_t* baseCounts,
    uint64_t k, const void* kmerTable, uint64_t kmerInfoStride, uint64_t isMarkerOffset, const uint8_t* readFlags,
    bool wantPacked, shasta_markers_result&);
void findMarkersFree(shasta_markers_result&);
// palindromic.hip: per read, an upper bound on the near-diagonal marker count of its self-alignment.
void palindromicScreen(Context&, uint64_t deltaThreshold, uint32_t* bound);
// tables.hip: the candidate / alignment table and the read graph's selection.
void pairTable(int device, const void* pairs, uint64_t stride, uint64_t count, uint64_t readCount, uint64_t* toc, uint32_t* values);
// Assembler::computeAlignmentTable for the alignments of the context's last borrowed aligner call; arrays of the context.
const shasta_alignment_data* borrowedAlignmentRows(Context&, uint64_t* count);
void alignmentTableOfLastCall(Context&, const uint64_t** toc, const uint32_t** values, uint64_t* valueCount);
void alignmentTableKeysBegin(Context&, uint64_t maxRows);          // (a borrowed aligner call begins)
void alignmentTableKeysOfBatch(Context&, const shasta_alignment_data* deviceRows, uint64_t count, uint64_t firstRow, hipStream_t stream);
void readGraphKeep(int device, const shasta_alignment_data* alignmentData, uint64_t count, uint64_t readCount, uint32_t maxAlignmentCount, uint8_t* keep);
void calibrateUnit(uint64_t bytes, int mode);
void hashWindowsUnit(const uint32_t* kmerIds, uint64_t n, uint64_t m, uint64_t iteration, uint64_t* out);
void bandedDpUnit(const uint32_t* k0, uint32_t nx, const uint32_t* k1, uint32_t ny, int32_t bandMin, int32_t bandMax,
    uint32_t* ordinals, uint64_t capacity, uint64_t* count, int32_t* score);

void bandedDpManyUnit(const uint32_t* kmerIds, uint64_t kmerCount, uint64_t taskCount,
    const uint64_t* begin0, const uint32_t* nx, const uint64_t* begin1, const uint32_t* ny, const int32_t* bandMin, const int32_t* bandMax,
    uint64_t* counts, int32_t* scores, uint32_t* ordinals, uint64_t capacity, double* seconds, uint64_t* cells);

}  // namespace shasta_mi355x
