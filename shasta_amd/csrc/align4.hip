// Align4 (Shasta alignment method 4) on MI355X (gfx950).  Replaces, for a batch of
// alignment candidates, Assembler::computeAlignmentsThreadFunction with
// alignMethod 4 (/root/reference/src/AssemblerAlign.cpp:308-496) ->
// Align4::align (src/Align4.cpp:30-166), AlignmentInfo::create
// (src/Alignment.cpp:67-113) and shasta::compress (src/compressAlignment.cpp:11-67).
//
//   K8/K9  align4CellsKernel   one workgroup per candidate: LDS hash join of the two
//                              marker sequences (replaces computeSortedMarkers + the merge
//                              join of createAlignmentMatrix, :195-267), per-cell entry
//                              counts in an LDS table (createCells :380-436), forward /
//                              backward reachability (:682-788) and 8-connected components
//                              (:792-868) by label propagation, one DP task per component.
//   K10    bandedDpKernel<C>   one wavefront per task: anti-diagonal banded overlap DP with
//                              each lane owning C adjacent diagonals (neighbour exchange =
//                              one __shfl per step), 2-bit trace packed with __ballot and
//                              streamed to HBM, wave-cooperative traceback through an LDS
//                              window (computeBandedAlignment :993-1088; the SeqAn call it
//                              wraps is restated -- tie policy as in oracle/banded_dp.hpp).
//   K11    select / finalize / compress kernels: best component (:126-147), filters
//                              (src/Align4.cpp:944-981, src/AssemblerAlign.cpp:439-472),
//                              AlignmentInfo, streak compression.
// Integer work throughout; no MFMA.  Bit-exactness notes: SURVEY.md Appendix A.2.
// This file: the shared types and constants, then the kernels (align4_cells.hpp K8/K9, align4_dp.hpp K10,
// align4_finish.hpp K11, align3.hpp for method 3), then the host side: batches, the class ladder of the cells
// stage, the DP launches, result assembly.
#include "context.hpp"

#include <algorithm>
#include <numeric>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <unordered_map>
#include <functional>
#include <thread>
#include <type_traits>

namespace shasta_mi355x {
namespace {

constexpr int NEG_SCORE = -(1 << 29);
constexpr int MATCH_SCORE = 6, MISMATCH_SCORE = -1, GAP_SCORE = -1;    // src/Align4.hpp:159-161

struct PairDesc { uint64_t begin0, begin1; uint32_t nx, ny; };
struct DpTask { uint32_t pair; int32_t bandMin, bandMax; uint32_t label; };
struct DpResult {
    uint64_t ordBegin;             // first (x,y) pair of this alignment in the ordinal scratch
    long long sumOffset;
    uint32_t markerCount, first0, first1, last0, last1;
    int32_t minOffset, maxOffset;
    uint32_t maxSkip, maxDrift;
    uint32_t passes;               // inner filters of src/Align4.cpp:944-981
    int32_t score;
    uint32_t compressedBytes;      // of this alignment in shasta::compress form (dpMetricsKernel)
};

struct DeviceOptions {
    uint32_t deltaX, deltaY;
    uint64_t minEntryCountPerCell, maxDistanceFromBoundary, minAlignedMarkerCount;
    double minAlignedFraction;
    uint64_t maxSkip, maxDrift, maxTrim, maxBand;
    uint32_t suppressContainments;
};

#include "align4_cells.hpp"      // K8/K9
#include "align4_dp.hpp"         // K10
#include "align4_sparse.hpp"     // K10s: the same alignment from the matches inside the band, where it is unique
#include "align4_chainwave.hpp"  // K10w: the chain recurrence with a wavefront per task, the task's hits in LDS
#include "align4_anchor.hpp"     // K10a: where it is not, the dense DP only between the matches every optimal alignment holds
#include "align4_finish.hpp"     // K11
#include "align3.hpp"

template<class T> T readDevice(const T* p, hipStream_t s)
{
    T v;
    HIP_CHECK(hipMemcpyAsync(&v, p, sizeof(T), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return v;
}

template<class T> T* mallocCopy(const std::vector<T>& v)
{
    T* p = static_cast<T*>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
    if(!p) throw std::bad_alloc();
    if(!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

DeviceOptions makeOptions(const shasta_align4_options& o)
{
    DeviceOptions d;
    // Aligner's constructor narrows deltaX/deltaY through int32_t (src/Align4.cpp:56-57).
    d.deltaX = uint32_t(int32_t(o.deltaX)); d.deltaY = uint32_t(int32_t(o.deltaY));
    if(d.deltaX == 0 || d.deltaY == 0) throw std::runtime_error("Align4: deltaX and deltaY must be positive.");
    d.minEntryCountPerCell = o.minEntryCountPerCell;
    d.maxDistanceFromBoundary = o.maxDistanceFromBoundary;
    d.minAlignedMarkerCount = o.minAlignedMarkerCount;
    d.minAlignedFraction = o.minAlignedFraction;
    d.maxSkip = o.maxSkip; d.maxDrift = o.maxDrift; d.maxTrim = o.maxTrim; d.maxBand = o.maxBand;
    // The banded DP kernels hold 1024 diagonals (src/AssemblerOptions.cpp: Align.maxBand defaults to 1000); a component that a
    // larger maxBand lets through (the reference only compares, src/Align4.cpp:929) runs in the wide DP (runWideTasks), up to
    // 65536 diagonals: a limit beyond that cannot be honoured and is refused here, loudly.
    if(o.maxBand > ALIGN3_HUGE_MAX_DIAGONALS) throw std::runtime_error("Align4: maxBand must not exceed 65536 (the widest band the wide DP holds).");
    d.suppressContainments = o.suppressContainments ? 1u : 0u;
    return d;
}

struct BatchScratch {
    DeviceBuffer<PairDesc> pairs;
    DeviceBuffer<shasta_oriented_read_pair> candidates;
    DeviceBuffer<DpTask> tasks;
    // What a batch starts from zero -- the best component and winner of every candidate, its flags, the task counters -- lies in
    // ONE block that one memset clears (five memsets in a row at the head of every batch until round 3): views into zeroBlock.
    template<class T> struct View { T* p = nullptr; T* data() const { return p; } };
    DeviceBuffer<uint8_t> zeroBlock;
    View<uint32_t> counters;                    // [0]=taskCount, [12]=tied candidates seen by winnerKernel, [13]=wide components
    void layoutZeroed(uint64_t n, hipStream_t stream)
    {
        const uint64_t bytes = 8 * n + 4 * n + 64 + n + n;
        zeroBlock.reserve(bytes + 16, stream);
        uint8_t* at = zeroBlock.data();
        pairBest.p = reinterpret_cast<unsigned long long*>(at); at += 8 * n;
        pairWinner.p = reinterpret_cast<uint32_t*>(at); at += 4 * n;
        counters.p = reinterpret_cast<uint32_t*>(at); at += 64;
        pairFlags.p = at; at += n;
        pairTie.p = at;
        HIP_CHECK(hipMemsetAsync(zeroBlock.data(), 0, bytes, stream));
    }
    DeviceBuffer<uint32_t> tieMembers, tieKeys, tieCounts;     // resolveComponentTies
    uint32_t sparseStateTasks = 0;       // of the batch in hand: the tasks sparseState speaks of (0: the sparse path did not run)
    bool streamsInLists = false;         // ... and whether the wave kernel leaves its alignments in shasta::compress form in their lists (compressWriteKernel copies them)
    bool pairsWanted = true;             // ... and whether anyone reads the aligned pairs of such a task (the caller asked for the ordinals): if not, the wave kernel does not write them
    DeviceBuffer<CellsChunk> tieChunks;
    View<uint8_t> pairFlags, pairTie;
    DeviceBuffer<uint8_t> status;
    View<unsigned long long> pairBest;
    View<uint32_t> pairWinner;
    DeviceBuffer<uint32_t> classLists, storedFlags, storedIndex, scanTemp32;
    DeviceBuffer<uint64_t> traceWords, ordCap, scanTemp64, trace, ordCounts, sizes;
    DeviceBuffer<uint32_t> ordScratch, ordOut;
    DeviceBuffer<DpResult> results;
    DeviceBuffer<shasta_alignment_data> rows, rowsOut;
    DeviceBuffer<uint64_t> compressedToc;
    DeviceBuffer<uint8_t> bytes, bigLog2;
    DeviceBuffer<uint32_t> pairList, bigScratch;
    DeviceBuffer<CellsChunk> chunks;
    DeviceBuffer<uint32_t> dpKeysA, dpKeysB, dpIdsA, dpIdsB;    // tasks sorted by (class, iterations)
    DeviceBuffer<uint64_t> bundleWords;         // where every bundle's trace begins
    DeviceBuffer<DpControl> dpControl;
    DeviceBuffer<DpEnd> ends;
    PinnedBuffer pinRows, pinToc, pinBytes, pinStatus, pinOrdToc, pinOrdinals;   // device-to-host staging
    PinnedBuffer pinTotals;                                                      // the few numbers a batch's host code waits for at its end, in ONE synchronisation (a copy to pageable memory blocks by itself)
    DeviceBuffer<uint64_t> bigOffsets;
    DeviceBuffer<PairDesc> dsPairs;             // align method 3, step 1: the down-sampled pairs
    DeviceBuffer<DpTask> tasks1;                //                         and their (unbanded) DP tasks
    DeviceBuffer<WideTask> wideTasks;           //                         pairs with more than 1024 diagonals
    DeviceBuffer<int32_t> hugeRows;             //                         the three anti-diagonals of the pairs with more than 8192 diagonals
    DeviceBuffer<WideEnd> wideEnds;
    DeviceBuffer<uint64_t> wideTrace, wideOrdBases;   // Align4 components of more than 1024 diagonals (runWideTasks)
    DeviceBuffer<uint32_t> hits, hitMeta, sparseSorted, sparseLinks, sparseAmbiguous, sparseInBand, denseFlags, densePositions;     // align4_sparse.hpp: the candidates' match lists, the tasks' ordered hits
    DeviceBuffer<uint64_t> hitBase;
    DeviceBuffer<uint8_t> sparseState;
    DeviceBuffer<uint32_t> chainWaveRetry;       // align4_chainwave.hpp: tasks too large for the class they were tried in, one list per further class
    DeviceBuffer<uint64_t> prepareKeysA, prepareKeysB;      // a batch's first chunk lists made on the device (align4_prepare.hpp)
    DeviceBuffer<uint32_t> prepareIdsA, prepareIdsB;
    DeviceBuffer<unsigned long long> prepareInfo;
    // (worker scratch only) all the buffers above, for raiseToMarks
    std::vector<SharedCapacityMember*> sharedBuffers;
    void raiseToMarks(hipStream_t stream) { for(SharedCapacityMember* m : sharedBuffers) m->raiseToMark(stream); }
    void scrambleAll(hipStream_t stream) { for(SharedCapacityMember* m : sharedBuffers) m->scramble(stream); }
};

// The scratch of a host worker: each of its buffers shares a high-water mark with the same buffer of the other workers.
std::shared_ptr<BatchScratch> makeWorkerScratch(SharedCapacities& shared)
{
    std::vector<SharedCapacityMember*> members;
    SharedCapacityBinding binding{&shared, &members, 0};
    struct Bound {
        Bound(SharedCapacityBinding* b) { sharedCapacityBinding = b; }
        ~Bound() { sharedCapacityBinding = nullptr; }
    } bound(&binding);
    std::shared_ptr<BatchScratch> scratch = std::make_shared<BatchScratch>();
    scratch->sharedBuffers = std::move(members);
    return scratch;
}

// A host worker's stream and sort workspace (two workers pipeline the batches of one call).
// `wide` is a side stream for the few wide-band DP tasks (one wavefront each, latency-bound): they
// overlap the narrow classes instead of occupying the GPU alone.
struct WorkStream { hipStream_t stream; RadixSortWorkspace* sortWs; hipStream_t wide; };

constexpr int CELLS_CLASSES = 6;
constexpr int CELLS_LONG = 4;          // the windowed class (align4CellsLongKernel): its own kernel instance, chunks of any sixteen candidates
constexpr int CELLS_LONG_BIG = 5;      // ... and its candidates that keep more cells than a wavefront's registers hold (align4CellsLongBigKernel): never a first choice, one candidate a chunk
#ifndef SHASTA_CELLS_NA0
#define SHASTA_CELLS_NA0 11
#endif
// Tabled read below 2048 / 4096 / 8192 / 8192 markers.  The fourth class (round 3) is for the pairs of two long reads, whose
// random matches touch more distinct cells than the third one's table holds (nx ny / 8192 of them at k = 10): 64 KB of cell
// region -- a byte grid up to nx + ny = 11 000, a packed table of 16 384 slots beyond -- and 256 kept cells, one workgroup per
// CU.  Until then those pairs (250 - 450 of a batch's 262 144 at 100 k reads) went to the kernel with its tables in HBM scratch:
// 1.6 ms per batch of 1024-thread workgroups that waited 95 % of their cycles (profiles/r02_pmc_100k_reads.json).
// The fifth class (round 6, CELLS_LONG) tables the SHORTER read of a candidate in windows of 2^13 markers: pairs of two reads beyond
// 8 192 markers (29 % of the candidates of the ultra-long shape, conf/Nanopore-UL-May2022.conf) and pairs whose cell indices do not
// fit the packed word of the others (nx + ny beyond 40 960 at deltaY = 10).
constexpr int CELLS_NA_LOG2[CELLS_CLASSES] = {SHASTA_CELLS_NA0, 12, 13, 13, 13, 13};
// Timing experiments compile other geometries (make EXTRA=-DSHASTA_CELLS_SC0=9 ...): the LDS a workgroup takes
// decides how many wavefronts a CU holds, and the cells kernels are bound by latency, not by instruction issue.
#ifndef SHASTA_CELLS_SC0
#define SHASTA_CELLS_SC0 11
#endif
#ifndef SHASTA_CELLS_SC1
#define SHASTA_CELLS_SC1 12
#endif
#ifndef SHASTA_CELLS_SC2
#define SHASTA_CELLS_SC2 12
#endif
#ifndef SHASTA_CELLS_ESTIMATE_SHIFT
#define SHASTA_CELLS_ESTIMATE_SHIFT 13
#endif
constexpr int CELLS_SC_LOG2[CELLS_CLASSES] = {SHASTA_CELLS_SC0, SHASTA_CELLS_SC1, SHASTA_CELLS_SC2, 14, 13, 13};
// Kept cells per candidate: 64 Q (more: the candidate climbs a class, finally to the HBM-scratch kernel).  Q = 2 everywhere:
// the kernel then needs 115 vector registers (4 wavefronts per SIMD) instead of 224 (2).
#ifndef SHASTA_CELLS_Q1
#define SHASTA_CELLS_Q1 2
#endif
#ifndef SHASTA_CELLS_Q2
#define SHASTA_CELLS_Q2 2
#endif
#ifndef SHASTA_CELLS_CHUNK_MAX
#define SHASTA_CELLS_CHUNK_MAX 24
#endif
constexpr int CELLS_Q[CELLS_CLASSES] = {2, SHASTA_CELLS_Q1, SHASTA_CELLS_Q2, 4, 4, 4};
constexpr uint32_t CELLS_CHUNK_MAX[CELLS_CLASSES] = {SHASTA_CELLS_CHUNK_MAX, SHASTA_CELLS_CHUNK_MAX, 16, 8, uint32_t(CELLS_LONG_WAVES), 1};
constexpr int ALIGN_DEFAULT_WORKERS = 6;                       // host workers (streams) that pipeline the batches of one call
#include "align4_prepare.hpp"    // the class of a candidate; a batch's first chunk lists made on the device

// Wavefronts of a chunk's workgroup: they stream one candidate at a time together, then take one kept-cell graph each.
#ifndef SHASTA_CELLS_WAVES
#define SHASTA_CELLS_WAVES 4
#endif
constexpr int CELLS_WAVES = SHASTA_CELLS_WAVES;
template<int Q>
void launchCellsChunksQ(Context& ctx, const WorkStream& ws, BatchScratch& b, int cls, const CellsChunk* chunks, uint32_t count,
    const DeviceOptions& opt, uint32_t magicX, uint32_t magicY, uint32_t taskCapacity, uint64_t kmerIdBytes, uint64_t candidateCount, const HitLists& hitLists)
{
    // (Fewer workgroups per CU -- 3 instead of 4, by padding the LDS request -- so that other workers' kernels find registers on
    // the same CU: 57 -> 66 ms solo and 214 -> 225 ms per step, scripts/gpu_r02_call29.sh.)
    const size_t bytes = cellsChunkLdsWords(CELLS_NA_LOG2[cls], CELLS_SC_LOG2[cls], Q, CELLS_WAVES) * sizeof(uint32_t);
    std::call_once(ctx.cellsLdsAttribute[Q == 2 ? 0 : 1], [] {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&align4CellsChunkKernel<Q>),
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    });
    MI355X_ASSERT(bytes <= 160 * 1024 - 1024);
    // Booked under the template instance that runs, the name a profiler shows.
    // Algorithmic bytes: 4 (nx + ny) per candidate, summed by the caller; work = candidates.
    const char* const name = Q == 2 ? "align4CellsChunkKernel<2, false>" : "align4CellsChunkKernel<4, false>";
    SHASTA_TIMED(ctx, name, ws.stream, kmerIdBytes, candidateCount,
        hipLaunchKernelGGL((align4CellsChunkKernel<Q>), dim3(count), dim3(WAVE * CELLS_WAVES), bytes, ws.stream,
            (const uint32_t*)ctx.kmerIds.data(), (const PairDesc*)b.pairs.data(), chunks, count, (const uint32_t*)b.pairList.data(),
            opt, magicX, magicY, b.tasks.data(), b.counters.data(), taskCapacity, b.pairFlags.data(), (uint32_t*)nullptr, (uint32_t*)nullptr, hitLists));
    HIP_CHECK(hipGetLastError());
}

// The windowed class: workgroups of sixteen wavefronts.  Algorithmic bytes as the other classes' (what the reference reads: 4 (nx + ny)
// per candidate; the kernel itself reads the stream once per window of the tabled read).
void launchCellsLong(Context& ctx, const WorkStream& ws, BatchScratch& b, const CellsChunk* chunks, uint32_t count,
    const DeviceOptions& opt, uint32_t magicX, uint32_t magicY, uint32_t taskCapacity, uint64_t kmerIdBytes, uint64_t candidateCount, const HitLists& hitLists, bool big)
{
    if(big) {
        // (the candidates with more kept cells than the class's graphs hold: one a workgroup, their kept cells behind the wavefronts' slots)
        const size_t bytes = (cellsChunkLdsWords(CELLS_NA_LOG2[CELLS_LONG_BIG], CELLS_SC_LOG2[CELLS_LONG_BIG], CELLS_Q[CELLS_LONG_BIG], CELLS_LONG_WAVES) + size_t(CELLS_BIG_KEPT)) * sizeof(uint32_t);
        std::call_once(ctx.cellsLdsAttribute[3], [] {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&align4CellsLongBigKernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        });
        MI355X_ASSERT(bytes <= 160 * 1024 - 1024);
        SHASTA_TIMED(ctx, "align4CellsLongBigKernel", ws.stream, kmerIdBytes, candidateCount,
            hipLaunchKernelGGL(align4CellsLongBigKernel, dim3(count), dim3(CELLS_LONG_THREADS), bytes, ws.stream,
                (const uint32_t*)ctx.kmerIds.data(), (const PairDesc*)b.pairs.data(), chunks, count, (const uint32_t*)b.pairList.data(),
                opt, magicX, magicY, b.tasks.data(), b.counters.data(), taskCapacity, b.pairFlags.data(), hitLists));
        HIP_CHECK(hipGetLastError());
        return;
    }
    // (SHASTA_MI355X_LONG_WAVES=<4 .. 16>: wavefronts of a workgroup of the windowed class -- a timing experiment; the LDS asked for stays that of sixteen)
    static const int wavesLong = [] { const char* e = std::getenv("SHASTA_MI355X_LONG_WAVES"); return e ? std::min(std::max(std::atoi(e), 4), CELLS_LONG_WAVES) : CELLS_LONG_WAVES; }();
    const size_t bytes = cellsChunkLdsWords(CELLS_NA_LOG2[CELLS_LONG], CELLS_SC_LOG2[CELLS_LONG], CELLS_Q[CELLS_LONG], CELLS_LONG_WAVES) * sizeof(uint32_t);
    std::call_once(ctx.cellsLdsAttribute[2], [] {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&align4CellsLongKernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    });
    MI355X_ASSERT(bytes <= 160 * 1024 - 1024);
    SHASTA_TIMED(ctx, "align4CellsLongKernel<false>", ws.stream, kmerIdBytes, candidateCount,
        hipLaunchKernelGGL((align4CellsLongKernel<false>), dim3(count), dim3(WAVE * wavesLong), bytes, ws.stream,
            (const uint32_t*)ctx.kmerIds.data(), (const PairDesc*)b.pairs.data(), chunks, count, (const uint32_t*)b.pairList.data(),
            opt, magicX, magicY, b.tasks.data(), b.counters.data(), taskCapacity, b.pairFlags.data(), (uint32_t*)nullptr, (uint32_t*)nullptr, hitLists));
    HIP_CHECK(hipGetLastError());
}

void launchCellsChunks(Context& ctx, const WorkStream& ws, BatchScratch& b, int cls, const CellsChunk* chunks, uint32_t count,
    const DeviceOptions& opt, uint32_t magicX, uint32_t magicY, uint32_t taskCapacity, uint64_t kmerIdBytes, uint64_t candidateCount, const HitLists& hitLists)
{
    if(count == 0) return;
    if(cls == CELLS_LONG || cls == CELLS_LONG_BIG) { launchCellsLong(ctx, ws, b, chunks, count, opt, magicX, magicY, taskCapacity, kmerIdBytes, candidateCount, hitLists, cls == CELLS_LONG_BIG); return; }
    if(CELLS_Q[cls] == 2) launchCellsChunksQ<2>(ctx, ws, b, cls, chunks, count, opt, magicX, magicY, taskCapacity, kmerIdBytes, candidateCount, hitLists);
    else launchCellsChunksQ<4>(ctx, ws, b, cls, chunks, count, opt, magicX, magicY, taskCapacity, kmerIdBytes, candidateCount, hitLists);
}

// What a DP runs on: the kmer-id array its pairs index, the pairs, the tasks.
struct DpInput { const uint32_t* kmerIds; const PairDesc* pairs; const DpTask* tasks; int tie; DpScores scores = {MATCH_SCORE, MISMATCH_SCORE, GAP_SCORE}; };
inline bool defaultScores(const DpScores& s) { return s.match == MATCH_SCORE && s.mismatch == MISMATCH_SCORE && s.gap == GAP_SCORE; }

// The tie policy of a call (align4_dp.hpp, DpTie): the build's DP_TIE_POLICY, unless SHASTA_MI355X_DP_TIE_POLICY names the one
// alternative compiled beside it (parity tests of the switch; read for every call: tests change it).
int dpTiePolicyOfCall()
{
    const char* e = std::getenv("SHASTA_MI355X_DP_TIE_POLICY");
    if(!e || !*e) return DP_TIE_POLICY;
    const int v = std::atoi(e);
    if(!dpTieCompiled(v))
        throw std::runtime_error("SHASTA_MI355X_DP_TIE_POLICY=" + std::string(e) + ": this build holds the tie policies " + std::to_string(DP_TIE_POLICY) +
            " (default), " + std::to_string(DP_TIE_ALTERNATIVE_A) + " and " + std::to_string(DP_TIE_ALTERNATIVE_B) + " (rebuild with -DSHASTA_DP_TIE_POLICY=<n> for another).");
    return v;
}

template<int G, int C>
void launchDpForward(const DpInput& in, hipStream_t stream, BatchScratch& b, const uint32_t* sortedIds, const DpClassLayout& layout, int cls)
{
    const uint32_t taskCount = layout.taskStart[cls + 1] - layout.taskStart[cls];
    const uint32_t bundleCount = layout.bundleStart[cls + 1] - layout.bundleStart[cls];
    if(taskCount == 0) return;
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(divUp(bundleCount, 4)), dim3(256), 0, stream,
            in.kmerIds, in.pairs, in.tasks,
            sortedIds + layout.taskStart[cls], taskCount,
            (const uint64_t*)(b.bundleWords.data() + layout.bundleStart[cls]), bundleCount,
            b.trace.data(), b.ends.data(), in.scores);
        HIP_CHECK(hipGetLastError());
    };
    if(!defaultScores(in.scores)) {
        if(in.tie != DP_TIE_POLICY) throw std::runtime_error("align method 3 with scores other than 6 / -1 / -1 is compiled for the default DP tie policy only.");
        launch(&bandedDpForwardKernel<G, C, DP_TIE_POLICY, true>);
    }
    else withDpTie(in.tie, [&](auto tag) { launch(&bandedDpForwardKernel<G, C, decltype(tag)::value>); });
}

// K10 for the taskCount tasks in b.tasks (pairs in b.pairs): fills b.results, b.ordScratch and
// b.pairBest.  Returns the number of DP cells (sum of nx * bandWidth); forwardSeconds gets the
// HIP-event time of the forward launches when evA/evB are given.
// Events that fork the wide-band classes of one batch's DP to the side stream and join them again.
struct DpEvents {
    hipEvent_t fork, join;
    void create() { HIP_CHECK(hipEventCreate(&fork)); HIP_CHECK(hipEventCreate(&join)); }
    void destroy() { (void)hipEventDestroy(fork); (void)hipEventDestroy(join); }
};
struct DpBatchStats { uint64_t cells[DP_CLASSES] = {0}, bytes[DP_CLASSES] = {0}; uint32_t tasks[DP_CLASSES] = {0}; };
// (the names a profiler shows for the default instances: tie policy DP_TIE_POLICY = 0, compile-time scores)
const char* const DP_FORWARD_NAMES[DP_CLASSES] = {"bandedDpForwardKernel<16, 2, 0, false>", "bandedDpForwardKernel<12, 4, 0, false>", "bandedDpForwardKernel<16, 4, 0, false>",
    "bandedDpForwardKernel<20, 4, 0, false>", "bandedDpForwardKernel<32, 4, 0, false>", "bandedDpForwardKernel<64, 4, 0, false>", "bandedDpForwardKernel<64, 8, 0, false>", "bandedDpForwardKernel<64, 16, 0, false>"};

// Forward half of K10 for taskCount tasks: sort by (band class, iterations), bundle, lay out the
// trace, run the forward kernel of every class.  Leaves b.trace / b.ends for a traceback kernel.
struct DpForwardState {
    const uint32_t* sortedIds;            // the tasks the dense kernels run: all of them, or (sparse path) those it did not certify
    uint32_t denseCount;                  // how many
    uint32_t taskStart[DP_CLASSES + 1];   // class c = tasks [taskStart[c], taskStart[c + 1]) of the sorted list
    uint32_t classCounts[DP_CLASSES];
    unsigned long long sums[2 + 2 * DP_CLASSES];   // [0] DP cells, [1] trace word bound, [2+c] cells of class c, [2+DP_CLASSES+c] bytes of class c
    uint64_t traceWords;                  // of the bundles' trace (2 bits per cell of the padded bands, once per bundle)
    uint64_t ordTotal;                    // ordinal pairs reserved for the tasks (the wide tasks' ranges follow)
};

// extraTasks / extraOrdinals: room behind the taskCount tasks for the wide tasks' results and aligned pairs.
// What the sparse path needs beside the tasks: the candidates' match lists (null: every task runs in the dense kernels).
struct SparseInput { const uint32_t* hits; const uint64_t* hitBase; const uint32_t* hitMeta; uint32_t maxOrdered = 0; };      // maxOrdered: the most markers a read the hits are ordered by has (the sort kernel's launches)
// SHASTA_MI355X_SPARSE_DP=0: the dense DP for every task (the A/B switch, and the second implementation the tests compare with).
bool sparseDpEnabled() { const char* e = std::getenv("SHASTA_MI355X_SPARSE_DP"); return !e || std::atoi(e) != 0; }
// SHASTA_MI355X_ANCHORED_DP=0: the tasks with several optimal chains go to the dense kernels whole (as before align4_anchor.hpp).
bool anchoredDpEnabled() { const char* e = std::getenv("SHASTA_MI355X_ANCHORED_DP"); return !e || std::atoi(e) != 0; }

__global__ void hitListNoneKernel(const uint32_t* __restrict__ list, uint32_t count, uint32_t* __restrict__ hitMeta)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if(k < count) hitMeta[list[k]] = HIT_LIST_NONE;
}
// SHASTA_MI355X_ANCHOR_BIG=0: no second launch of the anchor kernel (tasks with a rectangle beyond 4 096 cells go to the dense kernels, as before it).
bool anchorBigEnabled() { const char* e = std::getenv("SHASTA_MI355X_ANCHOR_BIG"); return !e || std::atoi(e) != 0; }
// SHASTA_MI355X_CHAIN_WAVE=0: every sorted task to sparseChainKernel (a lane per task), as before align4_chainwave.hpp.
bool chainWaveEnabled() { const char* e = std::getenv("SHASTA_MI355X_CHAIN_WAVE"); return !e || std::atoi(e) != 0; }

// SHASTA_MI355X_CHAIN_WAVE_SIDE=1: the two larger classes' launches on the worker's side stream, beside the first class's.
bool chainWaveSideStream() { const char* e = std::getenv("SHASTA_MI355X_CHAIN_WAVE_SIDE"); return e && std::atoi(e) != 0; }
// SHASTA_MI355X_CHAIN_WAVE_SORT=1: the wave kernel orders the hits itself (no sparseSortKernel); slower on the MI355X, kept for the A/B.
bool chainWaveOwnSort() { const char* e = std::getenv("SHASTA_MI355X_CHAIN_WAVE_SORT"); return e && std::atoi(e) != 0; }
// SHASTA_MI355X_CHAIN_WAVE_STREAM=0: the wave kernel leaves the shasta::compress form of its alignments to compressWriteKernel's pass over
// the aligned pairs (the form before: 1.5 GB read per launch there for 0.15 GB written).
bool chainWaveStream() { const char* e = std::getenv("SHASTA_MI355X_CHAIN_WAVE_STREAM"); return !e || std::atoi(e) != 0; }
// SHASTA_MI355X_CHAIN_WAVE_WIDE_D=1: D in 32 bits in every class (10 bytes of LDS per hit instead of 8: the form before, kept for the A/B).
bool chainWaveWideD() { const char* e = std::getenv("SHASTA_MI355X_CHAIN_WAVE_WIDE_D"); return e && std::atoi(e) != 0; }
template<int CLS, bool OWN_SORT, bool NARROW>
void launchChainWaveClassAs(hipStream_t stream, BatchScratch& b, const DpInput& in, uint32_t taskCount, const SparseInput& sparse, DpControl* control, const DeviceOptions& opt)
{
    constexpr uint32_t CAP = CHAIN_WAVE_CAPACITY[CLS];
    constexpr size_t ldsBytes = chainWaveLdsBytes(CAP, NARROW);        // (8 or 10 bytes per hit, 4 per window of 64)
    static_assert(CAP % 8 == 0, "the arrays behind the hits start at word boundaries");
    static_assert(ldsBytes <= 160u * 1024u, "a wavefront's hits in LDS");
    if(ldsBytes > 64u * 1024u) {
        // (more dynamic LDS than the default limit: the attribute once per device)
        static std::mutex mutex;
        static std::vector<int> done;           // (one per instantiation of this function template)
        int device = 0;
        HIP_CHECK(hipGetDevice(&device));
        std::lock_guard<std::mutex> lock(mutex);
        if(std::find(done.begin(), done.end(), device) == done.end()) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sparseChainWaveKernel<int(CAP), OWN_SORT, NARROW>), hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes)));
            done.push_back(device);
        }
    }
    // (SHASTA_MI355X_CHAIN_WAVE_SHARE=<percent>: the share of the wavefronts the LDS would let a CU hold that the launch asks for --
    // a timing experiment: what the kernel leaves of a CU's LDS is what the other workers' kernels can run in beside it)
    static const uint32_t share = [] { const char* e = std::getenv("SHASTA_MI355X_CHAIN_WAVE_SHARE"); return e ? uint32_t(std::min(std::max(std::atoi(e), 10), 100)) : 100u; }();
    const uint32_t grid = std::max<uint32_t>(256u, chainWaveGrid(CAP, NARROW) * share / 100u);
    hipLaunchKernelGGL((sparseChainWaveKernel<int(CAP), OWN_SORT, NARROW>), dim3(std::min<uint32_t>(grid, divUp(taskCount, CHAIN_WAVE_BLOCK))), dim3(64), ldsBytes, stream,
        in.pairs, in.tasks, taskCount, CLS, control,
        b.sparseSorted.data(), b.sparseInBand.data(), b.sparseState.data(), sparse.hits, sparse.hitBase, sparse.hitMeta,
        (const uint64_t*)b.ordCap.data(), b.ordScratch.data(), b.results.data(), b.sparseLinks.data(), b.ends.data(), b.sparseAmbiguous.data(), b.chainWaveRetry.data(), opt, b.pairBest.data(),
        (b.streamsInLists ? 1u : 0u) | (b.streamsInLists && !b.pairsWanted ? 2u : 0u));
    HIP_CHECK(hipGetLastError());
}
template<int CLS, bool OWN_SORT>
void launchChainWaveClass(hipStream_t stream, BatchScratch& b, const DpInput& in, uint32_t taskCount, const SparseInput& sparse, DpControl* control, const DeviceOptions& opt)
{
    if(chainWaveWideD()) launchChainWaveClassAs<CLS, OWN_SORT, false>(stream, b, in, taskCount, sparse, control, opt);
    else launchChainWaveClassAs<CLS, OWN_SORT, true>(stream, b, in, taskCount, sparse, control, opt);
}
// side + events: the launches of the two larger classes (13 % and 0.1 % of the tasks at 100 k reads, 8 and 1 wavefronts per CU) on the
// side stream beside the first class's, which they would otherwise follow: 2.2 + 1.9 + 2.0 ms one after the other (profiles/r05_call6).
void launchChainWave(hipStream_t stream, BatchScratch& b, const DpInput& in, uint32_t taskCount, const SparseInput& sparse, DpControl* control, const DeviceOptions& opt,
    hipStream_t side = nullptr, DpEvents* ev = nullptr)
{
    // (in the order of the classes: a launch lists the tasks that turned out too large for it for the next one)
    static_assert(CHAIN_WAVE_CLASSES == 6, "one launch per capacity class below");
    if(chainWaveOwnSort()) {
        launchChainWaveClass<0, true>(stream, b, in, taskCount, sparse, control, opt);
        launchChainWaveClass<1, true>(stream, b, in, taskCount, sparse, control, opt);
        launchChainWaveClass<2, true>(stream, b, in, taskCount, sparse, control, opt);
        launchChainWaveClass<3, true>(stream, b, in, taskCount, sparse, control, opt);
        launchChainWaveClass<4, true>(stream, b, in, taskCount, sparse, control, opt);
        launchChainWaveClass<5, true>(stream, b, in, taskCount, sparse, control, opt);
    } else if(side && ev && chainWaveSideStream()) {
        // (classes from the hits the sort kernel counted: no class lists tasks for another, so the launches need no order)
        HIP_CHECK(hipEventRecord(ev->fork, stream)); HIP_CHECK(hipStreamWaitEvent(side, ev->fork, 0));
        launchChainWaveClass<5, false>(side, b, in, taskCount, sparse, control, opt);
        launchChainWaveClass<4, false>(side, b, in, taskCount, sparse, control, opt);
        launchChainWaveClass<3, false>(side, b, in, taskCount, sparse, control, opt);
        launchChainWaveClass<2, false>(side, b, in, taskCount, sparse, control, opt);
        launchChainWaveClass<1, false>(side, b, in, taskCount, sparse, control, opt);
        HIP_CHECK(hipEventRecord(ev->join, side));
        launchChainWaveClass<0, false>(stream, b, in, taskCount, sparse, control, opt);
        HIP_CHECK(hipStreamWaitEvent(stream, ev->join, 0));
    } else {
        launchChainWaveClass<0, false>(stream, b, in, taskCount, sparse, control, opt);
        launchChainWaveClass<1, false>(stream, b, in, taskCount, sparse, control, opt);
        // (the listed classes: a batch without a task for one of them skips its launch -- the sort kernel's counts are not on the host, so the
        // launches go out whatever they hold; the few wavefronts of an empty class leave at once)
        launchChainWaveClass<2, false>(stream, b, in, taskCount, sparse, control, opt);
        launchChainWaveClass<3, false>(stream, b, in, taskCount, sparse, control, opt);
        launchChainWaveClass<4, false>(stream, b, in, taskCount, sparse, control, opt);
        launchChainWaveClass<5, false>(stream, b, in, taskCount, sparse, control, opt);
    }
}

DpForwardState runDpForward(const WorkStream& ws, BatchScratch& b, const DpInput& in, uint32_t taskCount, bool reserveOrdinals, DpEvents* ev, KernelTimers* timers,
    uint32_t extraTasks = 0, uint64_t extraOrdinals = 0, const SparseInput* sparse = nullptr, const DeviceOptions* metricsOptions = nullptr)
{
    hipStream_t stream = ws.stream;
    DpForwardState f;
    b.dpKeysA.reserve(taskCount, stream); b.dpKeysB.reserve(taskCount, stream);
    b.dpIdsA.reserve(taskCount, stream); b.dpIdsB.reserve(taskCount, stream);
    b.ordCap.reserve(uint64_t(taskCount) + 1, stream);
    b.results.reserve(uint64_t(taskCount) + extraTasks, stream); b.ends.reserve(taskCount, stream);
    b.dpControl.reserve(1, stream);
    DpControl* const control = b.dpControl.data();
    // What the preparation counts lives in one block (align4_dp.hpp, DpControl): one memset before, one copy to the host after.
    HIP_CHECK(hipMemsetAsync(control, 0, sizeof(DpControl), stream));
    KernelTimers::Span prepareSpan;
    if(timers) prepareSpan = timers->begin("DP task sizes, order by (class, length), bundles", stream);
    // The tasks in the order (class, length) by one counting pass (align4_dp.hpp): keys and bin counts, the bins' first positions, every
    // task to its place.  Their ranges of the ordinal scratch come from a cursor in the same kernel (b.ordCap = the ranges' starts).
    hipLaunchKernelGGL(dpSizeKernel, dim3(divUp(uint64_t(taskCount) + 1, 256)), dim3(256), 0, stream,
        in.tasks, in.pairs, taskCount, b.dpKeysA.data(), b.ordCap.data(), control, sparse ? 0 : 1);
    hipLaunchKernelGGL(dpBinScanKernel, dim3(1), dim3(1024), 0, stream, control);
    hipLaunchKernelGGL(dpScatterKernel, dim3(divUp(taskCount, 256)), dim3(256), 0, stream,
        (const uint32_t*)b.dpKeysA.data(), taskCount, control, b.dpKeysB.data(), b.dpIdsB.data());
    HIP_CHECK(hipGetLastError());
    const uint32_t* sortedKeys = b.dpKeysB.data();
    const uint32_t* sortedIds = b.dpIdsB.data();
    f.sortedIds = sortedIds; f.denseCount = taskCount;
    uint32_t* classCounts = f.classCounts;
    unsigned long long* sums = f.sums;
    size_t sortHandle = 0, chainHandle = 0, anchorHandle = 0, waveHandle = 0;
    bool anchored = false;
    if(sparse) {
        // K10s (align4_sparse.hpp): every task whose alignment is the unique optimal chain of the matches inside its band gets
        // it from those matches; the dense kernels below run what is left.  The tasks' ordered hits need room that depends on the
        // ordinal total: one synchronisation earlier than the dense path takes its own.
        MI355X_ASSERT(reserveOrdinals && metricsOptions);
        if(timers) (void)timers->end(prepareSpan, 16ULL * taskCount, taskCount);
        const uint64_t ordTotalEarly = readDevice(&control->ordCursor, stream);      // synchronises
        b.ordScratch.reserve(2 * (ordTotalEarly + extraOrdinals) + 2, stream);
        b.sparseSorted.reserve(2 * ordTotalEarly + 4, stream); b.sparseLinks.reserve(2 * ordTotalEarly + 4, stream);
        b.sparseAmbiguous.reserve(2 * uint64_t(taskCount), stream);          // (the chain kernels' list, and behind it the anchor kernel's list for its second launch)
        b.sparseInBand.reserve(taskCount, stream); b.sparseState.reserve(taskCount, stream);
        b.denseFlags.reserve(uint64_t(taskCount) + 1, stream); b.densePositions.reserve(uint64_t(taskCount) + 1, stream);
        b.scanTemp32.reserve(scanTempElements(uint64_t(taskCount) + 1), stream);
        const bool chainWave = chainWaveEnabled();
        KernelTimers::Span span;
        const bool ownSort = chainWave && chainWaveOwnSort();
        if(chainWave) b.chainWaveRetry.reserve(uint64_t(CHAIN_WAVE_CLASSES - 1) * taskCount, stream);       // (the listed classes' tasks; own-sort form: tasks handed on to the next class)
        if(!ownSort) {
        if(timers) span = timers->begin("sparseSortKernel", stream);
        hipLaunchKernelGGL((sparseSortKernel<4096, 0>), dim3(divUp(taskCount, 4)), dim3(256), 0, stream,
            in.pairs, in.tasks, taskCount, sparse->hits, sparse->hitBase, sparse->hitMeta, (const uint64_t*)b.ordCap.data(),
            b.sparseSorted.data(), b.sparseInBand.data(), b.sparseState.data(), control, chainWave, chainWave ? b.chainWaveRetry.data() : nullptr);
        hipLaunchKernelGGL((sparseSortKernel<8192, 4096>), dim3(divUp(taskCount, 4)), dim3(256), 0, stream,
            in.pairs, in.tasks, taskCount, sparse->hits, sparse->hitBase, sparse->hitMeta, (const uint64_t*)b.ordCap.data(),
            b.sparseSorted.data(), b.sparseInBand.data(), b.sparseState.data(), control, chainWave, chainWave ? b.chainWaveRetry.data() : nullptr);
        // (the windowed class's candidates: hits ordered by a read of up to 32 768 markers)
        if(sparse->maxOrdered > 8192) hipLaunchKernelGGL((sparseSortKernel<16384, 8192, 2>), dim3(divUp(taskCount, 2)), dim3(128), 0, stream,
            in.pairs, in.tasks, taskCount, sparse->hits, sparse->hitBase, sparse->hitMeta, (const uint64_t*)b.ordCap.data(),
            b.sparseSorted.data(), b.sparseInBand.data(), b.sparseState.data(), control, chainWave, chainWave ? b.chainWaveRetry.data() : nullptr);
        if(sparse->maxOrdered > 16384) hipLaunchKernelGGL((sparseSortKernel<int(SPARSE_MAX_STREAM), 16384, 1>), dim3(taskCount), dim3(64), 0, stream,
            in.pairs, in.tasks, taskCount, sparse->hits, sparse->hitBase, sparse->hitMeta, (const uint64_t*)b.ordCap.data(),
            b.sparseSorted.data(), b.sparseInBand.data(), b.sparseState.data(), control, chainWave, chainWave ? b.chainWaveRetry.data() : nullptr);
        HIP_CHECK(hipGetLastError());
        if(timers) sortHandle = timers->end(span, 0, taskCount);
        }
        if(chainWave) {
            // K10w (align4_chainwave.hpp): a wavefront per task, the task's hits in LDS; one launch of wavefronts per capacity class.
            if(timers) span = timers->begin("sparseChainWaveKernel", stream);
            launchChainWave(stream, b, in, taskCount, *sparse, control, *metricsOptions, ws.wide, ev);
            if(timers) waveHandle = timers->end(span, 0, taskCount);
        }
        if(!chainWave) {
        if(timers) span = timers->begin("sparseChainKernel", stream);
        hipLaunchKernelGGL(sparseChainKernel, dim3(divUp(taskCount, 64)), dim3(64), 0, stream,
            in.pairs, in.tasks, sortedIds, taskCount, b.sparseSorted.data(), (const uint32_t*)b.sparseInBand.data(), b.sparseState.data(), sparse->hitMeta,
            (const uint64_t*)b.ordCap.data(), b.ordScratch.data(), b.results.data(),
            b.sparseLinks.data(), b.ends.data(), b.sparseAmbiguous.data(), control, *metricsOptions, b.pairBest.data());
        HIP_CHECK(hipGetLastError());
        if(timers) chainHandle = timers->end(span, 0, taskCount);
        }
        // K10a (align4_anchor.hpp): the tasks with several optimal chains, the dense DP only where the chains differ.
        if(anchoredDpEnabled()) {
            anchored = true;
            if(timers) span = timers->begin("sparseAnchorKernel", stream);
            withDpTie(in.tie, [&](auto tag) {
                constexpr int TIE = decltype(tag)::value;
                auto launch = [&](auto big, uint32_t grid, size_t ldsBytes) {
                    constexpr bool BIG = decltype(big)::value;
                    hipLaunchKernelGGL((sparseAnchorKernel<TIE, BIG>), dim3(grid), dim3(64), ldsBytes, stream,
                        in.kmerIds, in.pairs, in.tasks, (const uint32_t*)b.sparseAmbiguous.data(), control,
                        (const uint32_t*)b.sparseSorted.data(), (const uint32_t*)b.sparseLinks.data(), (const uint32_t*)b.sparseInBand.data(), b.sparseState.data(),
                        sparse->hitMeta, (const DpEnd*)b.ends.data(), (const uint64_t*)b.ordCap.data(), b.ordScratch.data(), b.results.data(),
                        anchorBigEnabled() ? b.sparseAmbiguous.data() : nullptr, taskCount);
                };
                launch(std::false_type(), std::min<uint32_t>(ANCHOR_GRID, taskCount), sizeof(AnchorShared));
                if(anchorBigEnabled()) {
                    // The few tasks with a rectangle beyond the first launch's LDS: once more with 118 KB (dynamic LDS above the default limit:
                    // the attribute once per device).
                    static std::mutex mutex;
                    static std::vector<int> done;
                    int device = 0;
                    HIP_CHECK(hipGetDevice(&device));
                    {
                        std::lock_guard<std::mutex> lock(mutex);
                        if(std::find(done.begin(), done.end(), device * 16 + TIE) == done.end()) {
                            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sparseAnchorKernel<TIE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(AnchorSharedBig))));
                            done.push_back(device * 16 + TIE);
                        }
                    }
                    launch(std::true_type(), std::min<uint32_t>(256u, taskCount), sizeof(AnchorSharedBig));
                }
            });
            HIP_CHECK(hipGetLastError());
            if(timers) anchorHandle = timers->end(span, 0, taskCount);
        }
        if(timers) prepareSpan = timers->begin("DP task sizes, order by (class, length), bundles", stream);
        // The ordered list without the certified tasks (a compaction keeps the order), its class counts and sums.
        uint32_t* const denseKeys = b.dpKeysA.data();
        uint32_t* const denseIds = b.dpIdsA.data();
        hipLaunchKernelGGL(dpDenseFlagsKernel, dim3(divUp(uint64_t(taskCount) + 1, 256)), dim3(256), 0, stream, sortedIds, (const uint8_t*)b.sparseState.data(), taskCount, b.denseFlags.data());
        exclusiveScan<uint32_t>(b.denseFlags.data(), b.densePositions.data(), uint64_t(taskCount) + 1, b.scanTemp32.data(), stream);
        hipLaunchKernelGGL(dpDenseListKernel, dim3(divUp(taskCount, 256)), dim3(256), 0, stream,
            sortedKeys, sortedIds, (const uint32_t*)b.denseFlags.data(), (const uint32_t*)b.densePositions.data(), taskCount, in.tasks, in.pairs,
            denseKeys, denseIds, control->classCounts, control->sums);
        HIP_CHECK(hipGetLastError());
        sortedKeys = denseKeys; sortedIds = denseIds; f.sortedIds = denseIds;
    }
    // Where every bundle's trace lies (a thread per possible bundle: there are no more bundles than tasks), before the host knows
    // the class counts: one synchronisation for the counts, the ordinal total and the trace total.
    b.bundleWords.reserve(uint64_t(taskCount) + 1, stream);
    hipLaunchKernelGGL(dpBundleKernel, dim3(divUp(uint64_t(taskCount) + 1, 256)), dim3(256), 0, stream,
        sortedKeys, control, taskCount, b.bundleWords.data());
    HIP_CHECK(hipGetLastError());
    struct { unsigned long long sums[2 + 2 * DP_CLASSES], ordCursor, traceCursor; uint32_t classCounts[DP_CLASSES]; uint32_t ambiguousCount, pad; unsigned long long hitsListed, hitsInBand, ambiguousHits, giveUpCells[DP_GIVE_UP_REASONS]; uint32_t giveUpTasks[DP_GIVE_UP_REASONS]; } head;
    static_assert(sizeof(head) == DP_CONTROL_HEAD_BYTES && offsetof(DpControl, ordCursor) == sizeof(head.sums) && offsetof(DpControl, hitsListed) == 8 * (2 + 2 * DP_CLASSES + 2) + 4 * DP_CLASSES + 8, "DpControl's head");
    HIP_CHECK(hipMemcpyAsync(&head, control, sizeof(head), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    std::memcpy(classCounts, head.classCounts, sizeof(f.classCounts));
    std::memcpy(sums, head.sums, sizeof(f.sums));
    f.traceWords = head.traceCursor;
    const uint64_t ordTotal = head.ordCursor;
    const DpClassLayout layout = dpClassLayout(classCounts);
    MI355X_ASSERT(sparse ? layout.taskStart[DP_CLASSES] <= taskCount : layout.taskStart[DP_CLASSES] == taskCount);
    f.denseCount = layout.taskStart[DP_CLASSES];
    for(int c = 0; c <= DP_CLASSES; c++) f.taskStart[c] = layout.taskStart[c];
    if(timers) (void)timers->end(prepareSpan, 16ULL * taskCount, taskCount);
    if(timers && sparse) {
        // The sparse kernels' rows: the matches listed for the tasks' candidates are read (4 bytes each) and those inside the bands
        // written in order; the chain kernel reads those and writes a list word and a link word beside each; the anchor kernel reads both
        // for the tasks it walks and writes their aligned pairs (8 bytes: about one per match).  Work = matches inside the bands.
        if(!(chainWaveEnabled() && chainWaveOwnSort())) timers->amend(sortHandle, 4 * (head.hitsListed + head.hitsInBand), head.hitsInBand);
        // (with align4_chainwave.hpp on, nearly all tasks are the wave kernel's: it reads a hit once, 4 bytes, and writes a pair, 8)
        // (the wave kernel reads the listed matches twice, 4 bytes each, and writes a pair, 8 bytes, per match inside the bands)
        if(chainWaveEnabled()) { timers->amend(waveHandle, (chainWaveOwnSort() ? 8 * head.hitsListed + 8 * head.hitsInBand : 12 * head.hitsInBand), head.hitsInBand); }
        else timers->amend(chainHandle, 12 * head.hitsInBand, head.hitsInBand);
        if(anchored) timers->amend(anchorHandle, 16 * head.ambiguousHits, head.ambiguousHits);
        else if(head.ambiguousCount) timers->count(DP_GIVE_UP_NAMES[GIVE_UP_ANCHORS_OFF], head.ambiguousCount, 0, 0);
        // Why tasks went on to the dense kernels: rows without time (launches = tasks, work = their DP cells).
        for(int why = 0; why < DP_GIVE_UP_REASONS; why++) if(head.giveUpTasks[why]) timers->count(DP_GIVE_UP_NAMES[why], head.giveUpTasks[why], 0, head.giveUpCells[why]);
    }
    b.trace.reserve(f.traceWords + 64, stream);
    // (SHASTA_MI355X_TRACE_POISON=1: the trace filled with a byte pattern before the forward kernels write it -- a walk that reads a record
    // the forward kernel's stores have not delivered then reads that, not what an identical earlier call left at the same address)
    static const bool tracePoison = [] { const char* e = std::getenv("SHASTA_MI355X_TRACE_POISON"); return e && e[0] == '1'; }();
    if(tracePoison && f.traceWords) HIP_CHECK(hipMemsetAsync(b.trace.data(), 0xA5, (f.traceWords + 64) * sizeof(uint64_t), stream));
    f.ordTotal = ordTotal;
    if(reserveOrdinals) b.ordScratch.reserve(2 * (ordTotal + extraOrdinals) + 2, stream);

    // Wide bands (classes 5-7: few tasks, one wavefront each) go to the side stream, widest first;
    // the narrow classes run on the main stream meanwhile.
    // (SHASTA_MI355X_DP_FORK=0: every class on the main stream -- the A/B switch of the fork, for timing and for the search for order-dependent results;
    // align method 3 comes without events, i.e. without the fork, unless the switch is 1: alignRun.)
    const bool forkAllowed = [] { const char* e = std::getenv("SHASTA_MI355X_DP_FORK"); return !(e && e[0] == '0'); }();       // (read for every batch: tests and searches switch it)
    const bool fork = forkAllowed && ws.wide != nullptr && ev != nullptr && (classCounts[5] || classCounts[6] || classCounts[7]);
    hipStream_t wideStream = fork ? ws.wide : stream;
    if(fork) { HIP_CHECK(hipEventRecord(ev->fork, stream)); HIP_CHECK(hipStreamWaitEvent(ws.wide, ev->fork, 0)); }
    // Booked per class: algorithmic bytes 4 (nx + ny) per task, work = DP cells nx x bandWidth (dpSizeKernel's sums).
    auto timed = [&](int cls, hipStream_t st, auto launch) {
        if(!timers || classCounts[cls] == 0) { launch(st); return; }
        const KernelTimers::Span span = timers->begin(DP_FORWARD_NAMES[cls], st);
        launch(st);
        (void)timers->end(span, sums[2 + DP_CLASSES + cls], sums[2 + cls]);
    };
    timed(7, wideStream, [&](hipStream_t st) { launchDpForward<64, 16>(in, st, b, sortedIds, layout, 7); });
    timed(6, wideStream, [&](hipStream_t st) { launchDpForward<64, 8>(in, st, b, sortedIds, layout, 6); });
    timed(5, wideStream, [&](hipStream_t st) { launchDpForward<64, 4>(in, st, b, sortedIds, layout, 5); });
    if(fork) HIP_CHECK(hipEventRecord(ev->join, ws.wide));
    timed(2, stream, [&](hipStream_t st) { launchDpForward<16, 4>(in, st, b, sortedIds, layout, 2); });
    timed(1, stream, [&](hipStream_t st) { launchDpForward<12, 4>(in, st, b, sortedIds, layout, 1); });
    timed(3, stream, [&](hipStream_t st) { launchDpForward<20, 4>(in, st, b, sortedIds, layout, 3); });
    timed(4, stream, [&](hipStream_t st) { launchDpForward<32, 4>(in, st, b, sortedIds, layout, 4); });
    timed(0, stream, [&](hipStream_t st) { launchDpForward<16, 2>(in, st, b, sortedIds, layout, 0); });
    if(fork) HIP_CHECK(hipStreamWaitEvent(stream, ev->join, 0));
    return f;
}

// ---- component ties ------------------------------------------------------------------------------
// Align4 returns the FIRST alignment with the largest markerCount (src/Align4.cpp:126-147), in the order of the
// connected components of the active cells, and that order is the order of their union-find representatives
// (findActiveCellsConnectedComponents, :792-868: cell ids in (iY, iX) order, unions in the iteration order of a
// std::unordered_map keyed by MurmurHash64A of the coordinates, boost::disjoint_sets' union by rank, components in a
// std::map by representative).  Only a tie on markerCount between two components of a candidate can observe it.  For such
// candidates (none on the benchmark workload, a few on tandem repeats) the host repeats exactly that bookkeeping on the
// candidate's set of active cells -- the same container from the same standard library, the same hash, the same link rule --
// and picks the tied component whose representative is smallest.
uint64_t referenceMurmurHash64A(const void* key, int len, uint64_t seed)          // src/MurmurHash2.cpp:96-140
{
    const uint64_t m = 0xc6a4a7935bd1e995ULL;
    const int r = 47;
    uint64_t h = seed ^ (uint64_t(len) * m);
    const uint8_t* data = static_cast<const uint8_t*>(key);
    const uint8_t* const end = data + (len / 8) * 8;
    while(data != end) {
        uint64_t k;
        std::memcpy(&k, data, 8); data += 8;
        k *= m; k ^= k >> r; k *= m;
        h ^= k; h *= m;
    }
    switch(len & 7) {
        case 7: h ^= uint64_t(data[6]) << 48; [[fallthrough]];
        case 6: h ^= uint64_t(data[5]) << 40; [[fallthrough]];
        case 5: h ^= uint64_t(data[4]) << 32; [[fallthrough]];
        case 4: h ^= uint64_t(data[3]) << 24; [[fallthrough]];
        case 3: h ^= uint64_t(data[2]) << 16; [[fallthrough]];
        case 2: h ^= uint64_t(data[1]) << 8; [[fallthrough]];
        case 1: h ^= uint64_t(data[0]); h *= m;
    }
    h ^= h >> r; h *= m; h ^= h >> r;
    return h;
}
struct CoordinatesHash {                                                              // HashTuple<Coordinates>, src/hashArray.hpp:12-18
    size_t operator()(const std::pair<uint32_t, uint32_t>& v) const { return size_t(referenceMurmurHash64A(&v, sizeof(v), 15741)); }
};
// keys: the active cells (iY << 16 | iX), sorted ascending = the reference's cell ids.  Returns the representative of every cell.
std::vector<uint32_t> referenceComponentRepresentatives(const std::vector<uint32_t>& keys)
{
    const uint32_t n = uint32_t(keys.size());
    std::unordered_map<std::pair<uint32_t, uint32_t>, uint32_t, CoordinatesHash> activeCells;
    for(uint32_t id = 0; id < n; id++) activeCells.insert(std::make_pair(std::make_pair(keys[id] & 0xffffu, keys[id] >> 16), id));
    std::vector<uint32_t> rank(n, 0), parent(n);
    for(uint32_t i = 0; i < n; i++) parent[i] = i;
    auto findSet = [&](uint32_t v) {                                                 // find_with_full_path_compression
        uint32_t root = v;
        while(parent[root] != root) root = parent[root];
        while(parent[v] != root) { const uint32_t next = parent[v]; parent[v] = root; v = next; }
        return root;
    };
    auto unionSet = [&](uint32_t x, uint32_t y) {                                     // boost::disjoint_sets::union_set -> link_sets
        uint32_t i = findSet(x), j = findSet(y);
        if(i == j) return;
        if(rank[i] > rank[j]) parent[j] = i;
        else { parent[i] = j; if(rank[i] == rank[j]) ++rank[j]; }
    };
    for(const auto& p : activeCells) {
        const uint32_t iX0 = p.first.first, iY0 = p.first.second;
        for(int32_t dY = -1; dY <= 1; dY++) {
            const int32_t iY1 = int32_t(iY0) + dY;
            if(iY1 < 0) continue;
            for(int32_t dX = -1; dX <= 1; dX++) {
                if(dX == 0 && dY == 0) continue;
                const int32_t iX1 = int32_t(iX0) + dX;
                if(iX1 < 0) continue;
                const auto it = activeCells.find(std::make_pair(uint32_t(iX1), uint32_t(iY1)));
                if(it == activeCells.end()) continue;
                unionSet(p.second, it->second);
            }
        }
    }
    std::vector<uint32_t> representative(n);
    for(uint32_t i = 0; i < n; i++) representative[i] = findSet(i);
    return representative;
}

// After winnerKernel, before finalizeKernel.  pairClass[k]: the table class candidate k's cells were computed in (CELLS_CLASSES:
// the HBM-scratch kernel -- such a candidate keeps its tie flag).
void resolveComponentTies(Context& ctx, const WorkStream& ws, BatchScratch& b, uint32_t n, uint32_t taskCount, const std::vector<PairDesc>& hostPairs,
    const std::vector<int>& pairClass, const std::vector<uint8_t>& pairSlotsLog2, const std::vector<uint8_t>& pairNoGrid, const DeviceOptions& opt, uint32_t magicX, uint32_t magicY)
{
    hipStream_t stream = ws.stream;
    std::vector<uint8_t> tie(n);
    std::vector<uint32_t> winner(n);
    std::vector<unsigned long long> best(n);
    std::vector<DpTask> tasks(taskCount);
    std::vector<DpResult> results(taskCount);
    HIP_CHECK(hipMemcpyAsync(tie.data(), b.pairTie.data(), n, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(winner.data(), b.pairWinner.data(), n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(best.data(), b.pairBest.data(), n * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(tasks.data(), b.tasks.data(), taskCount * sizeof(DpTask), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(results.data(), b.results.data(), taskCount * sizeof(DpResult), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    // One chunk of one candidate per tied candidate, by class.
    std::vector<uint32_t> members;
    std::vector<CellsChunk> chunks[CELLS_CLASSES];
    std::vector<uint32_t> chunkPair[CELLS_CLASSES];
    std::vector<uint32_t> big;                                               // tied candidates of the HBM-scratch kernel
    for(uint32_t k = 0; k < n; k++) {
        if(tie[k] && (pairClass[k] == CELLS_CLASSES || pairClass[k] == CELLS_LONG_BIG) && pairSlotsLog2[k] >= 10) big.push_back(k);      // (the large graph has no dump form: the HBM-scratch kernel's)
        if(!tie[k] || pairClass[k] < 0 || pairClass[k] >= CELLS_LONG_BIG) continue;
        const int c = pairClass[k];
        const uint64_t capacity = 1ULL << CELLS_NA_LOG2[c];
        CellsChunk ch; ch.firstMember = uint32_t(members.size()); ch.count = 1;
        ch.swapped = (c == CELLS_LONG || hostPairs[k].nx < capacity) ? 0 : 1;     // (either read may be tabled: the cells are the same; the windowed class picks for itself)
        if(ch.swapped && hostPairs[k].ny >= capacity) continue;
        if(!pairNoGrid.empty() && pairNoGrid[k]) ch.swapped |= 2;             // (counted in the packed table the first time: again)
        ch.naLog2 = uint32_t(CELLS_NA_LOG2[c]); ch.scLog2 = uint32_t(CELLS_SC_LOG2[c]);
        members.push_back(k);
        chunks[c].push_back(ch); chunkPair[c].push_back(k);
    }
    if(std::getenv("SHASTA_MI355X_DEBUG")) std::fprintf(stderr, "ties: %zu + %zu candidates to look at again\n", members.size(), big.size());
    if(members.empty() && big.empty()) return;
    std::unordered_map<uint32_t, std::vector<uint32_t>> activeOf;              // candidate -> its active cells
    if(!members.empty()) {
        const uint32_t total = uint32_t(members.size());
        // A dump slot holds up to 64 Q keys, Q of the class's kernel instance.
        size_t keyWords = 0;
        for(int c = 0; c < CELLS_LONG_BIG; c++) keyWords += chunks[c].size() * size_t(64 * CELLS_Q[c]);
        b.tieMembers.reserve(total, stream); b.tieChunks.reserve(total, stream); b.tieKeys.reserve(keyWords, stream); b.tieCounts.reserve(total, stream);
        HIP_CHECK(hipMemcpyAsync(b.tieMembers.data(), members.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemsetAsync(b.tieCounts.data(), 0xff, total * sizeof(uint32_t), stream));
        uint32_t offset = 0;
        size_t wordOffset = 0;
        std::vector<uint32_t> order;                                         // candidate of dump slot s
        std::vector<size_t> slotWords;                                       // where slot s's keys start
        std::vector<uint32_t> slotCapacity;
        for(int c = 0; c < CELLS_LONG_BIG; c++) {
            if(chunks[c].empty()) continue;
            const uint32_t count = uint32_t(chunks[c].size());
            const uint32_t maxc = uint32_t(64 * CELLS_Q[c]);
            HIP_CHECK(hipMemcpyAsync(b.tieChunks.data() + offset, chunks[c].data(), count * sizeof(CellsChunk), hipMemcpyHostToDevice, stream));
            const int wavesOfClass = c == CELLS_LONG ? CELLS_LONG_WAVES : CELLS_WAVES;
            const size_t bytes = cellsChunkLdsWords(CELLS_NA_LOG2[c], CELLS_SC_LOG2[c], CELLS_Q[c], wavesOfClass) * sizeof(uint32_t);
            std::call_once(ctx.cellsDumpLdsAttribute, [] {
                HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&align4CellsChunkKernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
                HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&align4CellsChunkKernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
                HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&align4CellsLongKernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            });
            const auto kernel = c == CELLS_LONG ? &align4CellsLongKernel<true> : (CELLS_Q[c] == 2 ? &align4CellsChunkKernel<2, true> : &align4CellsChunkKernel<4, true>);
            hipLaunchKernelGGL(kernel, dim3(count), dim3(WAVE * wavesOfClass), bytes, stream,
                (const uint32_t*)ctx.kmerIds.data(), (const PairDesc*)b.pairs.data(), (const CellsChunk*)(b.tieChunks.data() + offset), count, (const uint32_t*)b.tieMembers.data(),
                opt, magicX, magicY, (DpTask*)nullptr, (uint32_t*)nullptr, 0u, (uint8_t*)nullptr, b.tieKeys.data() + wordOffset, b.tieCounts.data() + offset, HitLists{nullptr, nullptr, nullptr});
            HIP_CHECK(hipGetLastError());
            order.insert(order.end(), chunkPair[c].begin(), chunkPair[c].end());
            for(uint32_t q = 0; q < count; q++) { slotWords.push_back(wordOffset + size_t(q) * maxc); slotCapacity.push_back(maxc); }
            offset += count; wordOffset += size_t(count) * maxc;
        }
        std::vector<uint32_t> keys(keyWords), counts(total);
        HIP_CHECK(hipMemcpyAsync(keys.data(), b.tieKeys.data(), keys.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipMemcpyAsync(counts.data(), b.tieCounts.data(), total * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        for(uint32_t slot = 0; slot < total; slot++) {
            if(counts[slot] == 0xffffffffu || counts[slot] == 0 || counts[slot] > slotCapacity[slot]) continue;
            activeOf[order[slot]].assign(keys.begin() + slotWords[slot], keys.begin() + slotWords[slot] + counts[slot]);
        }
    }
    // The candidates of the HBM-scratch kernel, a few at a time (their key lists may be long).
    for(size_t begin = 0; begin < big.size(); ) {
        std::vector<uint64_t> scratchOffsets, keyOffsets;
        std::vector<uint8_t> log2s;
        uint64_t words = 0, keyWords = 0;
        size_t end = begin;
        while(end < big.size()) {
            const int l = pairSlotsLog2[big[end]];
            const uint64_t need = (9ULL << l) / 2, keyNeed = 1ULL << (l - 1);
            if(end > begin && (words + need > (1ULL << 31) || keyWords + keyNeed > (1ULL << 28))) break;
            scratchOffsets.push_back(words); keyOffsets.push_back(keyWords); log2s.push_back(uint8_t(l));
            words += need; keyWords += keyNeed; ++end;
        }
        const uint32_t count = uint32_t(end - begin);
        b.bigScratch.reserve(words, stream); b.bigOffsets.reserve(2 * size_t(count), stream); b.bigLog2.reserve(count, stream);
        b.tieMembers.reserve(count, stream); b.tieKeys.reserve(keyWords, stream); b.tieCounts.reserve(count, stream);
        HIP_CHECK(hipMemcpyAsync(b.tieMembers.data(), big.data() + begin, count * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(b.bigOffsets.data(), scratchOffsets.data(), count * 8ULL, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(b.bigOffsets.data() + count, keyOffsets.data(), count * 8ULL, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(b.bigLog2.data(), log2s.data(), count, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemsetAsync(b.tieCounts.data(), 0xff, count * sizeof(uint32_t), stream));
        hipLaunchKernelGGL((align4CellsKernel<true, true>), dim3(count), dim3(CELLS_THREADS), 0, stream,
            (const uint32_t*)ctx.kmerIds.data(), (const PairDesc*)b.pairs.data(), (const uint32_t*)b.tieMembers.data(), count, opt,
            (DpTask*)nullptr, (uint32_t*)nullptr, 0u, (uint8_t*)nullptr,
            b.bigScratch.data(), (const uint64_t*)b.bigOffsets.data(), (const uint8_t*)b.bigLog2.data(),
            b.tieKeys.data(), (const uint64_t*)(b.bigOffsets.data() + count), b.tieCounts.data());
        HIP_CHECK(hipGetLastError());
        std::vector<uint32_t> keys(keyWords), counts(count);
        HIP_CHECK(hipMemcpyAsync(keys.data(), b.tieKeys.data(), keyWords * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipMemcpyAsync(counts.data(), b.tieCounts.data(), count * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        for(uint32_t q = 0; q < count; q++) {
            if(counts[q] == 0xffffffffu || counts[q] == 0 || uint64_t(counts[q]) > (1ULL << (log2s[q] - 1))) continue;
            activeOf[big[begin + q]].assign(keys.begin() + keyOffsets[q], keys.begin() + keyOffsets[q] + counts[q]);
        }
        begin = end;
    }
    // Tasks by candidate (only the tied ones matter).
    std::unordered_map<uint32_t, std::vector<uint32_t>> tasksOf;
    for(uint32_t t = 0; t < taskCount; t++) if(tie[tasks[t].pair] && results[t].passes) tasksOf[tasks[t].pair].push_back(t);
    bool changed = false;
    for(auto& entry : activeOf) {
        const uint32_t k = entry.first;
        std::vector<uint32_t>& cells = entry.second;
        std::sort(cells.begin(), cells.end());
        const std::vector<uint32_t> representative = referenceComponentRepresentatives(cells);
        const uint32_t bestCount = uint32_t(best[k] >> 32);
        uint32_t chosenTask = 0xffffffffu, chosenRepresentative = 0xffffffffu;
        bool complete = true;
        for(uint32_t t : tasksOf[k]) {
            if(results[t].markerCount != bestCount) continue;
            const auto it = std::lower_bound(cells.begin(), cells.end(), tasks[t].label);
            if(it == cells.end() || *it != tasks[t].label) { complete = false; break; }
            const uint32_t r = representative[size_t(it - cells.begin())];
            if(r < chosenRepresentative) { chosenRepresentative = r; chosenTask = t; }
        }
        static const bool debug = std::getenv("SHASTA_MI355X_DEBUG") != nullptr;
        if(debug) std::fprintf(stderr, "ties: candidate %u: %zu active cells, %zu passing tasks, best %u markers, %s, task %u (was %u)\n", k, cells.size(),
            tasksOf[k].size(), bestCount, complete ? "complete" : "a task's seed cell is not among the active cells", chosenTask, winner[k]);
        if(!complete || chosenTask == 0xffffffffu) continue;
        winner[k] = chosenTask; tie[k] = 0; changed = true;
    }
    if(changed) {
        HIP_CHECK(hipMemcpyAsync(b.pairWinner.data(), winner.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(b.pairTie.data(), tie.data(), n, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));                             // the host vectors go out of scope
    }
}

// Components of more than 1024 diagonals (Align.maxBand beyond what the banded DP kernels hold; the reference only compares the
// band with maxBand, src/Align4.cpp:929): `wide` = their tasks, which sit at tasks[taskCount ...) on the device; each runs in
// align3WideDpKernel over its own diagonals (three anti-diagonals in LDS up to 8192 diagonals, in HBM scratch up to 65536), its
// path is walked by wideTracebackKernel, and it leaves results[taskCount + k] and its aligned pairs at ordBase + ... like any task.
void runWideTasks(Context& ctx, const WorkStream& ws, BatchScratch& b, const DpInput& in, uint32_t taskCount, const std::vector<DpTask>& wide,
    const std::vector<PairDesc>& hostPairs, uint64_t ordBase, uint64_t& dpCells)
{
    hipStream_t stream = ws.stream;
    const uint64_t traceWordBudget = 1ULL << 29;
    std::vector<uint64_t> ordBases(wide.size());
    for(size_t k = 0; k < wide.size(); k++) { ordBases[k] = ordBase; ordBase += std::min(hostPairs[wide[k].pair].nx, hostPairs[wide[k].pair].ny); }
    for(int pass = 0; pass < 2; pass++) {                       // 0: rows in LDS; 1: rows in HBM scratch
        for(size_t begin = 0; begin < wide.size(); ) {
            std::vector<WideTask> list;
            std::vector<uint64_t> bases;
            std::vector<uint32_t> resultIndex;
            uint64_t words = 0;
            uint32_t rowWords = 0;
            size_t end = begin;
            for(; end < wide.size(); end++) {
                const DpTask& t = wide[end];
                const PairDesc& pd = hostPairs[t.pair];
                const uint32_t width = uint32_t(t.bandMax - t.bandMin + 1);
                if(width > ALIGN3_HUGE_MAX_DIAGONALS) throw std::runtime_error("Align4: a component of more than 65536 diagonals (Align.maxBand beyond what the wide DP holds).");
                if((width > ALIGN3_WIDE_MAX_DIAGONALS) != (pass == 1)) continue;
                WideTask w; w.pair = t.pair; w.chunks = (width + 63) / 64; w.dMin = t.bandMin; w.width = width;
                const uint64_t need = 2ULL * (uint64_t(pd.nx) + pd.ny + 1) * w.chunks;
                if(!list.empty() && words + need > traceWordBudget) break;
                w.traceOffset = words; words += need;
                rowWords = std::max(rowWords, w.chunks * 64u);
                dpCells += uint64_t(pd.nx) * width;
                list.push_back(w); bases.push_back(ordBases[end]); resultIndex.push_back(uint32_t(end));
            }
            begin = end;
            if(list.empty()) continue;
            // (results must land at taskCount + index in `wide`: one launch per run of consecutive indices)
            const uint32_t count = uint32_t(list.size());
            b.wideTasks.reserve(count, stream); b.wideEnds.reserve(count, stream); b.wideTrace.reserve(words + 64, stream); b.wideOrdBases.reserve(count, stream);
            HIP_CHECK(hipMemcpyAsync(b.wideTasks.data(), list.data(), count * sizeof(WideTask), hipMemcpyHostToDevice, stream));
            HIP_CHECK(hipMemcpyAsync(b.wideOrdBases.data(), bases.data(), count * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
            std::call_once(ctx.wideDpLdsAttribute, [] {
                for(const int tie : {DP_TIE_POLICY, DP_TIE_ALTERNATIVE_A, DP_TIE_ALTERNATIVE_B}) withDpTie(tie, [](auto tag) {
                    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&align3WideDpKernel<false, decltype(tag)::value>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, int(3 * ALIGN3_WIDE_MAX_DIAGONALS * sizeof(int32_t)))); });
            });
            if(pass == 1) {
                b.hugeRows.reserve(size_t(count) * 3u * rowWords, stream);
                withDpTie(in.tie, [&](auto tag) {
                    hipLaunchKernelGGL((align3WideDpKernel<true, decltype(tag)::value>), dim3(count), dim3(256), 0, stream,
                        in.kmerIds, in.pairs, (const WideTask*)b.wideTasks.data(), count, rowWords, b.wideTrace.data(), b.wideEnds.data(), b.hugeRows.data(), in.scores); });
            } else {
                withDpTie(in.tie, [&](auto tag) {
                    hipLaunchKernelGGL((align3WideDpKernel<false, decltype(tag)::value>), dim3(count), dim3(64), 3 * size_t(rowWords) * sizeof(int32_t), stream,
                        in.kmerIds, in.pairs, (const WideTask*)b.wideTasks.data(), count, rowWords, b.wideTrace.data(), b.wideEnds.data(), (int32_t*)nullptr, in.scores); });
            }
            HIP_CHECK(hipGetLastError());
            // The tasks of this launch are not consecutive in `wide` when the two passes interleave: one traceback launch per task
            // index run keeps results[taskCount + index] right.
            for(uint32_t q = 0; q < count; ) {
                uint32_t r = q + 1;
                while(r < count && resultIndex[r] == resultIndex[r - 1] + 1) ++r;
                hipLaunchKernelGGL(wideTracebackKernel, dim3(r - q), dim3(64), 0, stream,
                    in.pairs, (const WideTask*)(b.wideTasks.data() + q), (const WideEnd*)(b.wideEnds.data() + q), r - q,
                    (const uint64_t*)b.wideTrace.data(), (const uint64_t*)(b.wideOrdBases.data() + q), b.ordScratch.data(), b.results.data(), taskCount + resultIndex[q]);
                HIP_CHECK(hipGetLastError());
                q = r;
            }
            HIP_CHECK(hipStreamSynchronize(stream));             // (the host lists go out of scope; the buffers are reused by the next group)
        }
    }
}

uint64_t runDpTasks(Context& ctx, const WorkStream& ws, BatchScratch& b, uint32_t taskCount, const DeviceOptions& opt,
    DpEvents* ev, DpBatchStats* stats, const DpScores* scores = nullptr, const std::vector<DpTask>* wide = nullptr, const std::vector<PairDesc>* hostPairs = nullptr,
    const SparseInput* sparse = nullptr)
{
    hipStream_t stream = ws.stream;
    DpInput in{ctx.kmerIds.data(), b.pairs.data(), b.tasks.data(), dpTiePolicyOfCall()};
    if(scores) in.scores = *scores;
    const uint32_t wideCount = wide ? uint32_t(wide->size()) : 0u;
    uint64_t wideOrdinals = 0;
    if(wideCount) for(const DpTask& t : *wide) wideOrdinals += std::min((*hostPairs)[t.pair].nx, (*hostPairs)[t.pair].ny);
    static const bool debug = std::getenv("SHASTA_MI355X_DEBUG") != nullptr;
    if(debug) {
        // Band widths of the batch's tasks, DP cells (nx x width) per width.
        std::vector<DpTask> hostTasks(taskCount);
        HIP_CHECK(hipMemcpyAsync(hostTasks.data(), b.tasks.data(), taskCount * sizeof(DpTask), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        uint32_t pairCount = 0;
        for(const DpTask& t : hostTasks) pairCount = std::max(pairCount, t.pair + 1);
        std::vector<PairDesc> hostPairs(pairCount);
        HIP_CHECK(hipMemcpyAsync(hostPairs.data(), b.pairs.data(), hostPairs.size() * sizeof(PairDesc), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        std::map<int32_t, std::pair<uint64_t, uint64_t>> histogram;
        for(const DpTask& t : hostTasks) {
            const int32_t w = t.bandMax - t.bandMin + 1;
            auto& h = histogram[w <= 300 ? w : ((w + 99) / 100) * 100];
            h.first += 1; h.second += uint64_t(hostPairs[t.pair].nx) * uint64_t(w);
        }
        std::fprintf(stderr, "dp: band width -> tasks, cells:");
        for(const auto& kv : histogram) std::fprintf(stderr, " %d: %llu, %.3g;", kv.first, (unsigned long long)kv.second.first, double(kv.second.second));
        std::fprintf(stderr, "\n");
    }
    DpForwardState f;
    std::memset(&f, 0, sizeof(f));
    b.sparseStateTasks = (taskCount && sparse && defaultScores(in.scores)) ? taskCount : 0u;
    b.streamsInLists = b.sparseStateTasks && chainWaveEnabled() && chainWaveStream();
    if(taskCount) f = runDpForward(ws, b, in, taskCount, true, ev, &ctx.timers, wideCount, wideOrdinals, (sparse && defaultScores(in.scores)) ? sparse : nullptr, &opt);
    else { b.results.reserve(wideCount, stream); b.ordScratch.reserve(2 * wideOrdinals + 2, stream); }
    // The traceback of every class in one launch (the list is sorted by class, then by ascending length; the kernel takes it from the end).
    // Booked: the trace it has to read = 2 bits per cell of the padded bands, once per bundle (the bundles' trace words as
    // dpBundleKernel laid them out; the tasks of a bundle walk the same records) -- work = tasks.
    if(f.denseCount) {
        SHASTA_TIMED(ctx, "dpTracebackKernel", stream, 8 * f.traceWords, f.denseCount,
            hipLaunchKernelGGL(dpTracebackKernel, dim3(divUp(f.denseCount, 256)), dim3(256), 0, stream,
                in.pairs, in.tasks, f.sortedIds, 0u, f.denseCount,
                (const DpEnd*)b.ends.data(), (const uint64_t*)b.trace.data(),
                (const uint64_t*)b.ordCap.data(), b.ordScratch.data(), b.results.data()));
        HIP_CHECK(hipGetLastError());
        // ... and the long paths, a wavefront each (the tasks of DP_TRACEBACK_LONG iterations and more: the others' wavefronts leave at once).
        SHASTA_TIMED(ctx, "dpTracebackWaveKernel", stream, 0, f.denseCount,
            hipLaunchKernelGGL(dpTracebackWaveKernel, dim3(f.denseCount), dim3(64), 0, stream,
                in.pairs, in.tasks, f.sortedIds, f.denseCount,
                (const DpEnd*)b.ends.data(), (const uint64_t*)b.trace.data(),
                (const uint64_t*)b.ordCap.data(), b.ordScratch.data(), b.results.data()));
        HIP_CHECK(hipGetLastError());
    }
    uint64_t wideCells = 0;
    if(wideCount) runWideTasks(ctx, ws, b, in, taskCount, *wide, *hostPairs, f.ordTotal, wideCells);
    // Booked: 8 bytes per aligned pair are read (unknown here: at most min(nx, ny) per task; the caller amends nothing) -- work = tasks.
    const uint32_t allTasks = taskCount + wideCount;
    SHASTA_TIMED(ctx, "dpMetricsKernel", stream, 0, allTasks,
        hipLaunchKernelGGL(dpMetricsKernel, dim3(divUp(uint64_t(allTasks) * WAVE, 256)), dim3(256), 0, stream,
            in.pairs, in.tasks, allTasks, (const uint32_t*)b.ordScratch.data(), b.results.data(), opt, b.pairBest.data(),
            b.sparseStateTasks ? (const uint8_t*)b.sparseState.data() : (const uint8_t*)nullptr, b.sparseStateTasks));
    HIP_CHECK(hipGetLastError());
    if(stats) for(int c = 0; c < DP_CLASSES; c++) { stats->cells[c] = f.sums[2 + c]; stats->bytes[c] = f.sums[2 + DP_CLASSES + c]; stats->tasks[c] = f.classCounts[c]; }
    return f.sums[0] + wideCells;
}

struct BatchOutput {
    std::vector<shasta_alignment_data> rows;
    std::vector<uint64_t> tocEnds, ordToc;
    std::vector<uint8_t> bytes;
    // The compressed bytes while they are still in the worker's pinned staging buffer and nowhere else (a batch that is placed
    // in the caller-visible array at once is copied from there: one host copy of a hundred megabytes instead of two); byteCount
    // is their number wherever they are.
    const uint8_t* stagedBytes = nullptr;
    uint64_t byteCount = 0;
    bool bytesInPlace = false;      // the device copied them straight to their place in the caller-visible array
    std::vector<uint32_t> ordinals;
    uint64_t dpCells = 0, kmerIdBytes = 0, alignedBytes = 0;
    DpBatchStats dpStats;
    bool hadTasks = false;
};

// Results of a "borrowed" call live here, in the context, and are reused by the next call: no
// half-gigabyte malloc / page-fault / munmap cycle per call.
// The caller-visible array of the compressed alignments of borrowed calls: page-locked, so that the device copies a batch's
// bytes straight to their place in it (no host copy at all); grows keeping its contents.
class PinnedBytes {
public:
    PinnedBytes() = default;
    PinnedBytes(const PinnedBytes&) = delete;
    PinnedBytes& operator=(const PinnedBytes&) = delete;
    ~PinnedBytes() { release(); }
    size_t size() const { return n; }
    uint8_t* data() const { return p; }
    bool pageLocked() const { return locked; }
    // (Where that much memory cannot be page-locked -- tens of gigabytes of results, a locked-memory limit -- the array is
    // ordinary memory and the batches come through the workers' staging buffers: one host copy each.)
    void resize(size_t m)
    {
        if(m <= n) return;
        void* q = nullptr;
        bool qLocked = true;
        // (SHASTA_MI355X_RESULTS_NOT_PAGE_LOCKED=1: as if page-locking had failed -- tests of the staging path.)
        static const bool never = [] { const char* e = std::getenv("SHASTA_MI355X_RESULTS_NOT_PAGE_LOCKED"); return e && e[0] == '1'; }();
        if(never || hipHostMalloc(&q, m, hipHostMallocDefault) != hipSuccess || !q) {
            (void)hipGetLastError();
            q = std::malloc(m);
            if(!q) throw std::bad_alloc();
            qLocked = false;
        }
        if(p) std::memcpy(q, p, n);
        release();
        p = static_cast<uint8_t*>(q); n = m; locked = qLocked;
    }
private:
    void release() { if(p) { if(locked) (void)hipHostFree(p); else std::free(p); } p = nullptr; n = 0; }
    uint8_t* p = nullptr;
    size_t n = 0;
    bool locked = true;
};

struct AlignStore {
    std::vector<BatchOutput> outputs;
    std::vector<shasta_alignment_data> rows;
    std::vector<uint64_t> compressedToc, ordinalsToc;
    PinnedBytes bytes;
    std::vector<uint8_t> status;
    std::vector<uint32_t> ordinals;
    uint64_t lastRowCount = 0;      // alignments stored by the last borrowed call (alignmentTableOfLastCall)
    bool valid = false;
};

// Align method 3: what alignOrientedReads3 needs besides the outer filters.
struct Align3Plan { uint32_t k, hashThreshold; int32_t bandExtend, maxBand; DpScores scores = {MATCH_SCORE, MISMATCH_SCORE, GAP_SCORE}; };

// The markers method 3 keeps in step 1 (src/AssemblerAlign3.cpp:66-82), for every oriented read:
// CSR of kmer ids and ordinals.  Built once per context and (k, threshold); dropped by setMarkers.
struct Downsampled {
    uint32_t k = 0, hashThreshold = 0;
    DeviceBuffer<uint64_t> toc, scanTemp;         // 2R+1
    DeviceBuffer<uint32_t> kmerIds, ordinals;
    std::vector<uint64_t> hostToc;
};

const Downsampled& ensureDownsampled(Context& ctx, const Align3Plan& plan)
{
    Downsampled* d = static_cast<Downsampled*>(ctx.downsampled.get());
    if(d && d->k == plan.k && d->hashThreshold == plan.hashThreshold) return *d;
    ctx.downsampled = std::make_shared<Downsampled>();
    d = static_cast<Downsampled*>(ctx.downsampled.get());
    d->k = plan.k; d->hashThreshold = plan.hashThreshold;
    hipStream_t stream = ctx.stream;
    const uint64_t orientedReadCount = 2 * ctx.readCount;
    d->toc.reserve(orientedReadCount + 1, stream);
    d->scanTemp.reserve(scanTempElements(orientedReadCount + 1), stream);
    const unsigned grid = divUp((orientedReadCount + 1) * WAVE, 256);
    hipLaunchKernelGGL(downsampleKernel<false>, dim3(grid), dim3(256), 0, stream,
        (const uint32_t*)ctx.kmerIds.data(), (const uint64_t*)ctx.toc.data(), orientedReadCount, plan.k, plan.hashThreshold,
        d->toc.data(), (const uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr);
    HIP_CHECK(hipGetLastError());
    exclusiveScan<uint64_t>(d->toc.data(), d->toc.data(), orientedReadCount + 1, d->scanTemp.data(), stream);
    d->hostToc.resize(orientedReadCount + 1);
    HIP_CHECK(hipMemcpyAsync(d->hostToc.data(), d->toc.data(), (orientedReadCount + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    const uint64_t total = d->hostToc[orientedReadCount];
    d->kmerIds.reserve(total + 1, stream); d->ordinals.reserve(total + 1, stream);
    hipLaunchKernelGGL(downsampleKernel<true>, dim3(grid), dim3(256), 0, stream,
        (const uint32_t*)ctx.kmerIds.data(), (const uint64_t*)ctx.toc.data(), orientedReadCount, plan.k, plan.hashThreshold,
        (uint64_t*)nullptr, (const uint64_t*)d->toc.data(), d->kmerIds.data(), d->ordinals.data());
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(stream));
    return *d;
}

// The widest step-1 matrix the register-resident DP classes hold (every diagonal of the down-sampled
// pair, -ny .. nx); wider ones, up to ALIGN3_WIDE_MAX_DIAGONALS, run in align3WideDpKernel.
constexpr uint32_t ALIGN3_MAX_STEP1_DIAGONALS = 1024;

// Both alignment methods.  m3 == nullptr: method 4 (DP tasks from the cells kernels); otherwise
// method 3 (DP tasks from the down-sampled step).  opt holds the outer filters either way.
void alignRun(Context& ctx, uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const DeviceOptions& opt, const Align3Plan* m3, bool wantOrdinals, shasta_align4_result& result, bool borrowed)
{
    std::memset(&result, 0, sizeof(result));
    const auto t0 = std::chrono::steady_clock::now();
    HIP_CHECK(hipSetDevice(ctx.device));
    const Downsampled* ds = m3 ? &ensureDownsampled(ctx, *m3) : nullptr;
    // Method 3 has no inner acceptance (src/Align4.cpp:944-981 belongs to method 4): its one DP per
    // pair is kept whenever it aligned anything, and only the outer filters apply.
    DeviceOptions dpOpt = opt;
    if(m3) {
        dpOpt.minAlignedMarkerCount = 0; dpOpt.minAlignedFraction = 0.;
        dpOpt.maxSkip = dpOpt.maxDrift = dpOpt.maxTrim = ~0ULL;
    }
    // Candidates per batch (SHASTA_MI355X_ALIGN_BATCH_LOG2, or SHASTA_MI355X_ALIGN_BATCH for a size that is not a power of
    // two, override it for timing experiments: 2^10 .. 2^20).
    static const uint64_t BATCH = []() -> uint64_t {
        if(const char* a = std::getenv("SHASTA_MI355X_ALIGN_BATCH")) return uint64_t(std::min(std::max(std::atol(a), 1024L), 1L << 20));
        const char* e = std::getenv("SHASTA_MI355X_ALIGN_BATCH_LOG2");
        const int l = e ? std::atoi(e) : 18;
        return 1ULL << std::min(std::max(l, 10), 20);
    }();
    // Where the batches begin: every BATCH candidates.  (Graded batches -- short first ones, because all workers prepare their
    // first batch together while the device waits, and shrinking last ones -- were measured against these and dropped: 171-174
    // against 167-171 ms per call, profiles/r02_call36_batch_schedule.log.)
    // ... or sooner, when the candidates' reads are long: a batch's scratch (match lists, DP tasks' ranges, traces) grows with the
    // markers of its candidates, not with their number, and 2^18 candidates of the ultra-long shape (15 000 markers a pair against
    // 3 000) asked the six workers for more than the 288 GB (round 5, bench.py --workload ul).  So a batch also ends at 4 000 markers
    // per candidate of a full batch: the 100 k-read workload's batches stay what they were.
    // The batches are cut from the END of the list backwards, so that the short one (what is left over) is the FIRST: a batch's
    // results go straight to their place in the caller-visible arrays only when every batch before it has said how much it puts out,
    // and a short LAST batch finished before its full-size predecessor had -- its 50-odd MB were then copied by one thread after the
    // device had gone idle, 5 to 11 ms of a 130 ms call (round 5, SHASTA_MI355X_LOG_HOST=1).
    std::vector<uint64_t> batchStart;
    {
        // (in blocks of 4 096 candidates counted from the end, their markers summed by four threads: one thread over two million
        // candidates was 3 ms at the head of every call)
        const uint64_t markerBudget = BATCH * 4000ULL;
        const uint64_t BLOCK = std::max<uint64_t>(1, std::min<uint64_t>(4096, BATCH / 8));      // (a batch is a whole number of blocks)
        const uint64_t blocks = (candidateCount + BLOCK - 1) / BLOCK;
        std::vector<uint64_t> blockMarkers(size_t(blocks), 0);
        auto sumBlocks = [&](uint64_t first, uint64_t stride) {
            for(uint64_t j = first; j < blocks; j += stride) {
                const uint64_t end = candidateCount - j * BLOCK, begin = end > BLOCK ? end - BLOCK : 0;
                uint64_t markers = 0;
                for(uint64_t k = begin; k < end; k++) {
                    const shasta_oriented_read_pair& c = candidates[k];
                    if(c.readIds[0] < ctx.readCount && c.readIds[1] < ctx.readCount) {       // (an invalid candidate is reported by the batch that meets it)
                        const uint64_t o0 = 2ULL * c.readIds[0], o1 = 2ULL * c.readIds[1];
                        markers += (ctx.hostToc[o0 + 1] - ctx.hostToc[o0]) + (ctx.hostToc[o1 + 1] - ctx.hostToc[o1]);
                    }
                }
                blockMarkers[size_t(j)] = markers;
            }
        };
        {
            const uint64_t threads = blocks >= 64 ? 4 : 1;
            std::vector<std::thread> others;
            for(uint64_t k = 1; k < threads; k++) others.emplace_back(sumBlocks, k, threads);
            sumBlocks(0, threads);
            for(std::thread& t : others) t.join();
        }
        uint64_t count = 0, markers = 0;
        batchStart.push_back(candidateCount);
        for(uint64_t j = 0; j < blocks; j++) {
            const uint64_t end = candidateCount - j * BLOCK, begin = end > BLOCK ? end - BLOCK : 0;
            if(count && (count + (end - begin) > BATCH || markers + blockMarkers[size_t(j)] > markerBudget)) { batchStart.push_back(end); count = 0; markers = 0; }
            count += end - begin; markers += blockMarkers[size_t(j)];
        }
        if(batchStart.back() != 0) batchStart.push_back(0);
        std::reverse(batchStart.begin(), batchStart.end());
    }
    const uint64_t batchCount = batchStart.size() - 1;

    if(borrowed && !ctx.alignStore) ctx.alignStore = std::make_shared<AlignStore>();
    // (borrowed calls: the alignment table's keys are made batch by batch from the rows on the device, shasta_mi355x_alignment_table)
    const bool tableKeys = borrowed;
    if(tableKeys) alignmentTableKeysBegin(ctx, candidateCount);
    AlignStore localStore;
    AlignStore& store = borrowed ? *static_cast<AlignStore*>(ctx.alignStore.get()) : localStore;
    std::vector<BatchOutput>& outputs = store.outputs;
    if(outputs.size() < batchCount) outputs.resize(batchCount);
    for(uint64_t k = 0; k < batchCount; k++) {
        BatchOutput& o = outputs[k];
        o.dpCells = o.kmerIdBytes = o.alignedBytes = 0; o.hadTasks = false;
        o.stagedBytes = nullptr; o.byteCount = 0; o.bytes.clear(); o.bytesInPlace = false;
        o.dpStats = DpBatchStats();
        o.ordToc.clear(); o.ordinals.clear();
    }
    std::vector<uint8_t>& outStatus = store.status;
    outStatus.resize(std::max<uint64_t>(1, candidateCount));

    // Several host workers, each with its own stream and grow-only scratch kept in the context, take
    // the batches in turn: one worker's host-side preparation, result copies and its launches that
    // cannot fill the device (a few long reads per batch) overlap the other workers' kernels.
    struct Worker {
        hipStream_t stream = nullptr;
        RadixSortWorkspace* sortWs = nullptr;
        BatchScratch* scratch = nullptr;
        hipStream_t wide = nullptr;
        DpEvents ev;
        std::vector<PairDesc> hostPairs;
        std::vector<uint64_t> hostToc64, hostHitBase;
        std::string error;
    };
    // SHASTA_MI355X_ALIGN_WORKERS overrides the number of workers (for timing experiments; 1 .. ALIGN_MAX_WORKERS).
    const int configuredWorkers = [] {                 // read at every call: bench.py times one pass with a single worker
        const char* e = std::getenv("SHASTA_MI355X_ALIGN_WORKERS");
        const int n = e ? std::atoi(e) : ALIGN_DEFAULT_WORKERS;
        return std::min(std::max(n, 1), int(Context::ALIGN_MAX_WORKERS));
    }();
    const int workerCount = int(std::min<uint64_t>(uint64_t(configuredWorkers), std::max<uint64_t>(1, batchCount)));
    std::vector<Worker> workers;
    workers.resize(size_t(workerCount));
    for(int k = 0; k < workerCount; k++) {
        if(!ctx.alignScratch[k]) ctx.alignScratch[k] = makeWorkerScratch(ctx.alignCapacities);
        if(k > 0 && !ctx.workerStream[k]) HIP_CHECK(hipStreamCreateWithFlags(&ctx.workerStream[k], hipStreamNonBlocking));
        workers[k].stream = k == 0 ? ctx.stream : ctx.workerStream[k];
        workers[k].sortWs = k == 0 ? &ctx.sortWs : &ctx.workerSortWs[k];
        workers[k].scratch = static_cast<BatchScratch*>(ctx.alignScratch[k].get());
        if(!ctx.wideStream[k]) HIP_CHECK(hipStreamCreateWithFlags(&ctx.wideStream[k], hipStreamNonBlocking));
        workers[k].wide = ctx.wideStream[k];
        workers[k].ev.fork = ctx.alignEvent(3 + 2 * size_t(k)); workers[k].ev.join = ctx.alignEvent(4 + 2 * size_t(k));       // (the context's, kept from call to call)
    }
    hipEvent_t evBegin, evEnd, evOther;
    evBegin = ctx.alignEvent(0); evEnd = ctx.alignEvent(1); evOther = ctx.alignEvent(2);
    HIP_CHECK(hipEventRecord(evBegin, ctx.stream));
    // SHASTA_MI355X_LOG_HOST=1: where the call's wall clock goes on the host (ms since entry), on stderr.
    static const bool logHost = std::getenv("SHASTA_MI355X_LOG_HOST") != nullptr;
    auto hostMs = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const double msAtBegin = hostMs();

    // A batch says how much it will put out (stored alignments, compressed bytes, ordinal pairs) as soon as the device has
    // told it, before the results are written and copied: its place in the caller-visible arrays is the sum over the batches
    // before it, which have said theirs by then (defined with the placement, below).
    std::function<uint8_t*(uint64_t, uint64_t, uint64_t, uint64_t)> publishSizes;      // -> where the batch's bytes go in the caller-visible array, if that is known and has room
    auto processBatch = [&](Worker& w, uint64_t batchIndex) {
        hipStream_t stream = w.stream;
        const WorkStream ws{w.stream, w.sortWs, w.wide};
        BatchScratch& b = *w.scratch;
        BatchOutput& out = outputs[batchIndex];
        std::vector<PairDesc>& hostPairs = w.hostPairs;
        std::vector<uint64_t>& hostToc64 = w.hostToc64;
        const uint64_t batchBegin = batchStart[batchIndex];
        const uint32_t n = uint32_t(batchStart[batchIndex + 1] - batchBegin);
        hostPairs.resize(n);
        // Method 4 with the sparse path: room for every candidate's list of matches (align4_sparse.hpp), laid out here.
        const bool listHits = !m3 && sparseDpEnabled();
        std::vector<uint64_t>& hostHitBase = w.hostHitBase;
        if(listHits) { hostHitBase.resize(uint64_t(n) + 1); hostHitBase[0] = 0; }
        uint32_t maxOrdered = 0;
        const CellsClassRule listClassRule = cellsClassRule(opt, ctx.matchShift);
        for(uint32_t k = 0; k < n; k++) {
            const shasta_oriented_read_pair& c = candidates[batchBegin + k];
            if(!(c.readIds[0] < c.readIds[1]) || c.readIds[1] >= ctx.readCount) {
                throw std::runtime_error("Align4: invalid alignment candidate (need readId0 < readId1 < readCount).");
            }
            const uint64_t o0 = 2ULL * c.readIds[0];                                  // strand 0, src/AssemblerAlign.cpp:382
            const uint64_t o1 = 2ULL * c.readIds[1] + (c.isSameStrand ? 0 : 1);      // :383
            PairDesc pd;
            pd.begin0 = ctx.hostToc[o0]; pd.begin1 = ctx.hostToc[o1];
            const uint64_t nx = ctx.hostToc[o0 + 1] - pd.begin0, ny = ctx.hostToc[o1 + 1] - pd.begin1;
            // (the DP's sort key holds (nx + ny) / 2 iterations in 24 bits, its biased scores i + j < 2^25: align4_dp.hpp)
            if(nx + ny >= (1ULL << 25) - 4) throw std::runtime_error("Align4: a candidate's two reads have 2^25 - 4 markers or more between them (not supported).");
            // (align method 3 with its own scores: biased scores stay inside 32 bits while magnitude x length < 2^28)
            if(m3 && !defaultScores(m3->scores)) {
                const uint64_t magnitude = uint64_t(std::abs(m3->scores.match)) + uint64_t(std::abs(m3->scores.mismatch)) + uint64_t(std::abs(m3->scores.gap));
                if(magnitude * (nx + ny + 2) >= (1ULL << 28)) throw std::runtime_error("Align3: the scores times the length of a candidate's reads exceed the DP kernels' 32-bit range (not supported).");
            }
            pd.nx = uint32_t(nx); pd.ny = uint32_t(ny);
            hostPairs[k] = pd;
            out.kmerIdBytes += 4 * (nx + ny);
            // (no room for a candidate whose cells the HBM-scratch kernel computes -- both reads beyond the LDS tables: it lists no matches)
            if(listHits) {
                const CellsChoice choice = cellsChoice(listClassRule, pd.nx, pd.ny);
                hostHitBase[k + 1] = hostHitBase[k] + ((nx < 65535 && ny < 65535 && choice.cls < CELLS_CLASSES) ? hitListCapacity(pd.nx, pd.ny, ctx.matchShift) : 0u);
                if(choice.cls == CELLS_LONG || choice.cls == CELLS_LONG_BIG) maxOrdered = std::max(maxOrdered, std::min(pd.nx, pd.ny));       // (the windowed class orders by the shorter read; the others by a read below 8 192 markers)
            }
        }
        // Room for the DP tasks of the batch; the stage runs again with the exact count if it is short.
        // SHASTA_MI355X_INITIAL_TASKS overrides the first guess (tests use it to force the second run).
        b.raiseToMarks(stream);          // what other workers' batches needed so far: grown to now, not in the middle of the batch
        // (SHASTA_MI355X_SCRAMBLE=1, a test switch: every scratch buffer of the worker overwritten with pseudo-random data before the batch touches
        // it -- a kernel that reads what its own batch has not written then answers differently from call to call)
        if(const char* e = std::getenv("SHASTA_MI355X_SCRAMBLE")) if(e[0] == '1') b.scrambleAll(stream);
        uint32_t taskCapacity = 8 * n + 1024;
        if(const char* e = std::getenv("SHASTA_MI355X_INITIAL_TASKS")) taskCapacity = uint32_t(std::max(1L, std::atol(e)));
        b.pairs.reserve(n, stream); b.candidates.reserve(n, stream); b.tasks.reserve(taskCapacity, stream);
        b.status.reserve(n, stream);
        b.layoutZeroed(n, stream);       // counters, pairFlags, pairTie, pairBest, pairWinner: one block, one memset
        b.storedFlags.reserve(n + 1, stream); b.storedIndex.reserve(n + 1, stream);
        b.scanTemp32.reserve(scanTempElements(uint64_t(n) + 1), stream);
        b.ordCounts.reserve(n + 1, stream); b.sizes.reserve(n + 1, stream);
        b.rows.reserve(n, stream); b.rowsOut.reserve(n, stream); b.compressedToc.reserve(n + 1, stream);
        HIP_CHECK(hipMemcpyAsync(b.pairs.data(), hostPairs.data(), n * sizeof(PairDesc), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(b.candidates.data(), candidates + batchBegin, n * sizeof(shasta_oriented_read_pair), hipMemcpyHostToDevice, stream));
        HitLists hitLists{nullptr, nullptr, nullptr};
        if(listHits) {
            b.hits.reserve(hostHitBase[n] + 1, stream); b.hitBase.reserve(uint64_t(n) + 1, stream); b.hitMeta.reserve(n, stream);
            HIP_CHECK(hipMemcpyAsync(b.hitBase.data(), hostHitBase.data(), (uint64_t(n) + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
            HIP_CHECK(hipMemsetAsync(b.hitMeta.data(), 0xff, n * sizeof(uint32_t), stream));        // HIT_LIST_NONE until a chunk kernel lists the candidate's matches
            hitLists = HitLists{b.hits.data(), b.hitBase.data(), b.hitMeta.data()};
        }

        // SHASTA_MI355X_DEBUG: where a batch spends its time on the host's clock (the kernels of other workers run meanwhile).
        static const bool debugPhases = std::getenv("SHASTA_MI355X_DEBUG") != nullptr;
        const auto phaseClock = [] { return std::chrono::steady_clock::now(); };
        const auto phaseStart = phaseClock();
        auto phaseMs = [&](std::chrono::steady_clock::time_point from) { return std::chrono::duration<double, std::milli>(phaseClock() - from).count(); };
        double phaseCells = 0., phaseDp = 0., phaseFinish = 0.;
        uint32_t taskCount = 0, wideCount = 0;
        bool tieCounterWanted = false;                 // (the batch has DP tasks of method 4: two components of a candidate may tie)
        uint32_t tieTasks = 0;
        std::vector<int> pairClass;                    // method 4: the table class of every candidate's cells (CELLS_CLASSES: HBM scratch)
        std::vector<uint8_t> pairSlotsLog2;            //           ... and the table size the HBM-scratch kernel last ran it with
        std::vector<uint8_t> pairNoGrid;               //           ... and whether its cells were counted in the packed table because a byte of its grid overflowed
        uint32_t cellsMagicX = 0, cellsMagicY = 0;
        uint32_t hostCounters[16];
        for(;;) {
        bool flagsAndCountersCurrent = false;
        if(m3) {
            // Method 3, step 1: every diagonal of the down-sampled pair, then the band of step 2.
            std::vector<PairDesc> dsPairs(n);
            std::vector<DpTask> tasks1;
            std::vector<WideTask> wide, huge;
            std::vector<uint8_t> hostFlags(n, 0);
            tasks1.reserve(n);
            for(uint32_t k = 0; k < n; k++) {
                const shasta_oriented_read_pair& c = candidates[batchBegin + k];
                const uint64_t o0 = 2ULL * c.readIds[0], o1 = 2ULL * c.readIds[1] + (c.isSameStrand ? 0 : 1);
                PairDesc pd;
                pd.begin0 = ds->hostToc[o0]; pd.begin1 = ds->hostToc[o1];
                pd.nx = uint32_t(ds->hostToc[o0 + 1] - pd.begin0); pd.ny = uint32_t(ds->hostToc[o1 + 1] - pd.begin1);
                dsPairs[k] = pd;
                if(pd.nx == 0 || pd.ny == 0) continue;                           // empty alignment, src/AssemblerAlign3.cpp:101-107
                const uint64_t diagonals = uint64_t(pd.nx) + pd.ny + 1;
                if(diagonals <= ALIGN3_MAX_STEP1_DIAGONALS) {
                    DpTask t; t.pair = k; t.bandMin = -int32_t(pd.ny); t.bandMax = int32_t(pd.nx); t.label = 0;
                    tasks1.push_back(t);
                } else if(diagonals <= ALIGN3_HUGE_MAX_DIAGONALS) {
                    WideTask t; t.pair = k; t.chunks = uint32_t((diagonals + 63) / 64); t.traceOffset = 0; t.dMin = -int32_t(pd.ny); t.width = uint32_t(diagonals);
                    (diagonals <= ALIGN3_WIDE_MAX_DIAGONALS ? wide : huge).push_back(t);
                } else {
                    hostFlags[k] = PAIR_TOO_LONG;
                }
            }
            const uint32_t taskCount1 = uint32_t(tasks1.size());
            b.dsPairs.reserve(n, stream); b.tasks1.reserve(std::max<uint32_t>(1, taskCount1), stream);
            HIP_CHECK(hipMemcpyAsync(b.dsPairs.data(), dsPairs.data(), n * sizeof(PairDesc), hipMemcpyHostToDevice, stream));
            HIP_CHECK(hipMemcpyAsync(b.pairFlags.data(), hostFlags.data(), n, hipMemcpyHostToDevice, stream));
            if(taskCount1) {
                HIP_CHECK(hipMemcpyAsync(b.tasks1.data(), tasks1.data(), taskCount1 * sizeof(DpTask), hipMemcpyHostToDevice, stream));
                const DpInput in{ds->kmerIds.data(), b.dsPairs.data(), b.tasks1.data(), dpTiePolicyOfCall(), m3->scores};
                const DpForwardState f = runDpForward(ws, b, in, taskCount1, false, nullptr, nullptr);
                out.dpCells += f.sums[0];
                hipLaunchKernelGGL(align3BandKernel<false>, dim3(divUp(taskCount1, 256)), dim3(256), 0, stream,
                    in.pairs, (const PairDesc*)b.pairs.data(), in.tasks, f.sortedIds, taskCount1,
                    (const DpEnd*)b.ends.data(), (const WideTask*)nullptr, (const WideEnd*)nullptr,
                    (const uint64_t*)b.trace.data(), (const uint32_t*)ds->ordinals.data(),
                    m3->bandExtend, m3->maxBand, b.tasks.data(), b.counters.data(), taskCapacity, uint32_t(CELLS_WIDE_COUNTER));
                HIP_CHECK(hipGetLastError());
            }
            // The long pairs, a few gigabytes of trace at a time.
            const uint64_t traceWordBudget = 1ULL << 29;
            auto runWide = [&](std::vector<WideTask>& wide, bool hugeRows) {
            for(size_t begin = 0; begin < wide.size(); ) {
                size_t end = begin;
                uint64_t words = 0;
                uint32_t rowWords = 0;
                while(end < wide.size()) {
                    const PairDesc& pd = dsPairs[wide[end].pair];
                    const uint64_t need = 2ULL * (uint64_t(pd.nx) + pd.ny + 1) * wide[end].chunks;
                    if(end > begin && words + need > traceWordBudget) break;
                    wide[end].traceOffset = words;
                    words += need;
                    rowWords = std::max(rowWords, wide[end].chunks * 64u);
                    out.dpCells += uint64_t(pd.nx) * (uint64_t(pd.nx) + pd.ny + 1);
                    ++end;
                }
                const uint32_t count = uint32_t(end - begin);
                b.wideTasks.reserve(count, stream); b.wideEnds.reserve(count, stream); b.trace.reserve(words + 64, stream);
                HIP_CHECK(hipMemcpyAsync(b.wideTasks.data(), wide.data() + begin, count * sizeof(WideTask), hipMemcpyHostToDevice, stream));
                const size_t ldsBytes = 3 * size_t(rowWords) * sizeof(int32_t);
                std::call_once(ctx.wideDpLdsAttribute, [] {
                    for(const int tie : {DP_TIE_POLICY, DP_TIE_ALTERNATIVE_A, DP_TIE_ALTERNATIVE_B}) withDpTie(tie, [](auto tag) {
                        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&align3WideDpKernel<false, decltype(tag)::value>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, int(3 * ALIGN3_WIDE_MAX_DIAGONALS * sizeof(int32_t)))); });
                });
                if(hugeRows) {
                    b.hugeRows.reserve(size_t(count) * 3u * rowWords, stream);
                    withDpTie(dpTiePolicyOfCall(), [&](auto tag) {
                        hipLaunchKernelGGL((align3WideDpKernel<true, decltype(tag)::value>), dim3(count), dim3(256), 0, stream,
                            (const uint32_t*)ds->kmerIds.data(), (const PairDesc*)b.dsPairs.data(), (const WideTask*)b.wideTasks.data(), count, rowWords,
                            b.trace.data(), b.wideEnds.data(), b.hugeRows.data(), m3->scores); });
                } else {
                    withDpTie(dpTiePolicyOfCall(), [&](auto tag) {
                        hipLaunchKernelGGL((align3WideDpKernel<false, decltype(tag)::value>), dim3(count), dim3(64), ldsBytes, stream,
                            (const uint32_t*)ds->kmerIds.data(), (const PairDesc*)b.dsPairs.data(), (const WideTask*)b.wideTasks.data(), count, rowWords,
                            b.trace.data(), b.wideEnds.data(), (int32_t*)nullptr, m3->scores); });
                }
                HIP_CHECK(hipGetLastError());
                hipLaunchKernelGGL(align3BandKernel<true>, dim3(divUp(count, 256)), dim3(256), 0, stream,
                    (const PairDesc*)b.dsPairs.data(), (const PairDesc*)b.pairs.data(), (const DpTask*)nullptr, (const uint32_t*)nullptr, count,
                    (const DpEnd*)nullptr, (const WideTask*)b.wideTasks.data(), (const WideEnd*)b.wideEnds.data(),
                    (const uint64_t*)b.trace.data(), (const uint32_t*)ds->ordinals.data(),
                    m3->bandExtend, m3->maxBand, b.tasks.data(), b.counters.data(), taskCapacity, uint32_t(CELLS_WIDE_COUNTER));
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipStreamSynchronize(stream));
                begin = end;
            }
            };
            runWide(wide, false);
            runWide(huge, true);
            HIP_CHECK(hipStreamSynchronize(stream));      // the host vectors above are done with
        } else
        // K8/K9.  Chunks of candidates sharing read 0 run in LDS (three table-size classes);
        // whatever overflows its tables climbs one class, and finally runs with tables in HBM
        // scratch (align4CellsKernel<true>, retried with 8x the slots on overflow).
        {
            std::vector<uint32_t> bigList;
            std::vector<uint8_t> bigLog2, hostFlags(n);
            auto estimateLog2 = [&](uint32_t k) {
                const uint64_t nx = hostPairs[k].nx, ny = hostPairs[k].ny;
                const uint64_t cells = nx * ny / 4096 + (nx + ny) / 8 + 1024;
                int l = 13;
                while((1ULL << l) < 2 * cells && l < 24) ++l;
                return uint8_t(l);
            };
            const CellsClassRule classRule = cellsClassRule(opt, ctx.matchShift);            // (the class of a candidate: align4_prepare.hpp)
            pairClass.assign(n, -1); pairSlotsLog2.assign(n, 0); pairNoGrid.assign(n, 0);
            std::vector<CellsChunk> classChunks[CELLS_CLASSES];
            std::vector<uint32_t> members;                             // candidate indices, chunk after chunk
            members.reserve(n);
            auto addChunk = [&](const uint32_t* list, uint32_t count, bool swapped, int c, bool noGrid = false) {
                CellsChunk ch; ch.firstMember = uint32_t(members.size()); ch.count = uint16_t(count); ch.swapped = uint16_t((swapped ? 1 : 0) | (noGrid ? 2 : 0));
                ch.naLog2 = uint32_t(CELLS_NA_LOG2[c]); ch.scLog2 = uint32_t(CELLS_SC_LOG2[c]);
                members.insert(members.end(), list, list + count);
                classChunks[c].push_back(ch);
            };
            // deltaX, deltaY >= 2 here (packedOk fails for 1 x anything >= 2046... and d = 1 gives magic 2^32): guard.
            const uint32_t magicX = uint32_t(std::min<uint64_t>((1ULL << 32) / opt.deltaX + 1, 0xffffffffULL));
            const uint32_t magicY = uint32_t(std::min<uint64_t>((1ULL << 32) / opt.deltaY + 1, 0xffffffffULL));
            cellsMagicX = magicX; cellsMagicY = magicY;
            // Every candidate is a member once, plus once per class it climbs to after an overflow, plus once more if a byte of its grid overflowed.
            const uint64_t memberCapacity = uint64_t(CELLS_CLASSES + 1) * n + 16;       // (its first class, the classes it climbs to, and one repeat in the packed table)
            b.pairList.reserve(memberCapacity, stream);
            size_t membersUploaded = 0;
            // The first round's member list and chunk lists are made by kernels on the batch's stream and its cells kernels launched from
            // them at once; what the host needs for the later rounds (every candidate's class, the HBM-scratch list) it
            // computes while they run.  Same chunks, same order within a class (align4_prepare.hpp).
            // SHASTA_MI355X_DEVICE_BATCH_PREP=0: the host loop + sort + walk that it replaced (kept as the cross-check of the lists; on the
            // MI355X the device's lists take a step's aligner calls from 172 to 165 ms, profiles/r03_device_batch_prep_ab.log).
            const bool devicePrepare = [] { const char* e = std::getenv("SHASTA_MI355X_DEVICE_BATCH_PREP"); return !e || std::atoi(e) != 0; }();      // (read for every batch: tests switch it)
            bool firstRoundLaunched = false, firstRoundAny = false;
            if(devicePrepare) {
                int tabledBits = 1;
                while(tabledBits < 58 && ((2 * ctx.readCount) >> tabledBits) != 0) ++tabledBits;
                b.prepareKeysA.reserve(n, stream); b.prepareKeysB.reserve(n, stream); b.prepareIdsA.reserve(n, stream); b.prepareIdsB.reserve(n, stream);
                b.prepareInfo.reserve(CELLS_PREPARE_INFO, stream); b.chunks.reserve(n, stream);
                HIP_CHECK(hipMemsetAsync(b.prepareInfo.data(), 0, CELLS_PREPARE_INFO * sizeof(unsigned long long), stream));
                hipLaunchKernelGGL(cellsClassKeysKernel, dim3(divUp(n, 256)), dim3(256), 0, stream,
                    (const PairDesc*)b.pairs.data(), (const shasta_oriented_read_pair*)b.candidates.data(), n, classRule, tabledBits, b.prepareKeysA.data(), b.prepareIdsA.data(), b.prepareInfo.data());
                const bool inB = radixSort<uint64_t, uint32_t, true>(b.prepareKeysA.data(), b.prepareKeysB.data(), b.prepareIdsA.data(), b.prepareIdsB.data(),
                    n, tabledBits + 1 + CELLS_CLASS_BITS, *ws.sortWs, stream);
                const uint64_t* sortedKeys = inB ? b.prepareKeysB.data() : b.prepareKeysA.data();
                const uint32_t* sortedIds = inB ? b.prepareIdsB.data() : b.prepareIdsA.data();
                hipLaunchKernelGGL(cellsChunkHeadsKernel, dim3(divUp(uint64_t(n) + 1, 256)), dim3(256), 0, stream, sortedKeys, n, tabledBits, b.storedFlags.data());
                exclusiveScan<uint32_t>(b.storedFlags.data(), b.storedIndex.data(), uint64_t(n) + 1, b.scanTemp32.data(), stream);
                hipLaunchKernelGGL(cellsChunkWriteKernel, dim3(divUp(uint64_t(n) + 1, 256)), dim3(256), 0, stream,
                    sortedKeys, (const uint32_t*)b.storedIndex.data(), n, tabledBits, b.chunks.data(), b.prepareInfo.data());
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipMemcpyAsync(b.pairList.data(), sortedIds, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
                unsigned long long info[CELLS_PREPARE_INFO];
                HIP_CHECK(hipMemcpyAsync(info, b.prepareInfo.data(), sizeof(info), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
                // The classes with the large tables (one or two workgroups per CU: 140 KB of LDS in the last) hold few candidates and wait long
                // for whole CUs while other workers' kernels run -- 0.37 ms alone, 3.9 ms in a step for the last class's 275 candidates
                // (profiles/r05_call19.log).  SHASTA_MI355X_CELLS_SIDE_FROM=<class> puts the classes from that one on on the worker's side
                // stream, beside the first classes' launches; measured: 137.7 / 136.7 / 137.4 ms per step with 3 / 2 / none
                // (profiles/r05_call20.log) -- no difference, so none is the default (4).
                static const int sideFrom = [] { const char* e = std::getenv("SHASTA_MI355X_CELLS_SIDE_FROM"); return e ? std::atoi(e) : CELLS_CLASSES; }();
                bool onSide = false;
                for(int pass = 0; pass < 2; pass++) {
                    for(int c = 0; c < CELLS_CLASSES; c++) {
                        const bool side = ws.wide != nullptr && c >= sideFrom;
                        if(side != (pass == 0)) continue;                 // (the side stream's launches first: they wait longest)
                        const uint32_t count = uint32_t(info[c + 1] - info[c]);
                        if(count == 0) continue;
                        firstRoundAny = true;
                        const WorkStream where{side ? ws.wide : ws.stream, ws.sortWs, ws.wide};
                        launchCellsChunks(ctx, where, b, c, b.chunks.data() + info[c], count, opt, magicX, magicY, taskCapacity, info[CELLS_INFO_BYTES + c], info[CELLS_INFO_CANDIDATES + c], hitLists);
                        onSide = onSide || side;
                    }
                    if(pass == 0 && onSide) HIP_CHECK(hipEventRecord(w.ev.join, ws.wide));
                }
                if(onSide) HIP_CHECK(hipStreamWaitEvent(stream, w.ev.join, 0));
                firstRoundLaunched = true;
                members.assign(n, 0);                       // (positions 0 .. n-1 of the device's list: later rounds append after them)
                membersUploaded = n;
                for(uint32_t q = 0; q < n; q++) {
                    const int c = cellsChoice(classRule, hostPairs[q].nx, hostPairs[q].ny).cls;
                    pairClass[q] = c;
                    if(c == CELLS_CLASSES) { bigList.push_back(q); bigLog2.push_back(estimateLog2(q)); }
                    if(c == CELLS_LONG_BIG) pairSlotsLog2[q] = estimateLog2(q);       // (a tie of its components is looked at by the HBM-scratch kernel's dump form)
                }
                MI355X_ASSERT(uint64_t(n) - info[CELLS_INFO_FIRST_BIG] == bigList.size());
                static const bool debugPrepare = std::getenv("SHASTA_MI355X_DEBUG") != nullptr;
                if(debugPrepare) for(int c = 0; c < CELLS_CLASSES; c++) std::fprintf(stderr, "cells: lists made on the device: class %d: %llu chunks, %llu candidates\n", c, info[c + 1] - info[c], info[CELLS_INFO_CANDIDATES + c]);
            } else
            // Every candidate tables whichever of its two reads lands in the smaller class (ties: read
            // 0) and is grouped with the other candidates that table the same oriented read.
            {
                struct Keyed { uint64_t tabled; uint32_t pair; uint8_t cls, swapped; };
                std::vector<Keyed> keyed;
                keyed.reserve(n);
                for(uint32_t q = 0; q < n; q++) {
                    const PairDesc& pd = hostPairs[q];
                    const CellsChoice choice = cellsChoice(classRule, pd.nx, pd.ny);
                    const bool sw = choice.swapped;
                    const int c = choice.cls;
                    pairClass[q] = c;
                    if(c == CELLS_LONG_BIG) pairSlotsLog2[q] = estimateLog2(q);
                    if(c == CELLS_CLASSES) { bigList.push_back(q); bigLog2.push_back(estimateLog2(q)); continue; }
                    Keyed kd; kd.tabled = sw ? pd.begin1 : pd.begin0; kd.pair = q; kd.cls = uint8_t(c); kd.swapped = sw ? 1 : 0;
                    if(c == CELLS_LONG || c == CELLS_LONG_BIG) { kd.tabled = 0; kd.swapped = 0; }      // (the windowed class: one group, chunks of any sixteen candidates)
                    keyed.push_back(kd);
                }
                // Order: (swapped, tabled read, class, candidate).  The candidates arrive in ascending order, so a STABLE sort on the
                // composite key (swapped | tabled | class) is that order: an LSD radix sort with byte digits over the bits the
                // key uses -- 4 ms for a batch where std::sort with a four-field comparison took 21 (a sixth of a step for a
                // worker, and at the start of a call every worker does this before its first kernel can start).
                {
                    uint64_t maxTabled = 0;
                    for(const Keyed& kd : keyed) maxTabled = std::max(maxTabled, kd.tabled);
                    int tabledBits = 1;
                    while(tabledBits < 58 && (maxTabled >> tabledBits) != 0) ++tabledBits;
                    const int keyBits = 1 + tabledBits + 3;                     // swapped | tabled | class (CELLS_CLASSES <= 8)
                    static_assert(CELLS_CLASSES <= 8, "three bits of class in the sort key");
                    std::vector<uint64_t> keyA(keyed.size()), keyB(keyed.size());
                    std::vector<Keyed> other(keyed.size());
                    for(size_t k = 0; k < keyed.size(); k++) keyA[k] = (uint64_t(keyed[k].swapped) << (tabledBits + 3)) | (keyed[k].tabled << 3) | uint64_t(keyed[k].cls);
                    for(int shift = 0; shift < keyBits; shift += 8) {
                        size_t counts[257] = {0};
                        for(uint64_t key : keyA) ++counts[((key >> shift) & 0xff) + 1];
                        for(int d = 0; d < 256; d++) counts[d + 1] += counts[d];
                        for(size_t k = 0; k < keyed.size(); k++) {
                            const size_t to = counts[(keyA[k] >> shift) & 0xff]++;
                            keyB[to] = keyA[k]; other[to] = keyed[k];
                        }
                        keyA.swap(keyB); keyed.swap(other);
                    }
                }
                std::vector<uint32_t> list;
                for(size_t k = 0; k < keyed.size(); ) {
                    size_t e = k + 1;
                    const int c = keyed[k].cls;
                    while(e < keyed.size() && e - k < CELLS_CHUNK_MAX[c] && keyed[e].swapped == keyed[k].swapped &&
                        keyed[e].tabled == keyed[k].tabled && keyed[e].cls == keyed[k].cls) ++e;
                    list.clear();
                    for(size_t q = k; q < e; q++) list.push_back(keyed[q].pair);
                    addChunk(list.data(), uint32_t(list.size()), keyed[k].swapped != 0, c);
                    k = e;
                }
            }
            static const bool debug = std::getenv("SHASTA_MI355X_DEBUG") != nullptr;
            if(debug) {
                for(int c = 0; c < CELLS_CLASSES; c++) {
                    uint64_t pairsIn = 0, sw = 0;
                    for(const CellsChunk& ch : classChunks[c]) { pairsIn += ch.count; sw += ch.swapped; }
                    std::fprintf(stderr, "cells: class %d: %zu chunks, %llu candidates, %llu swapped\n", c,
                        classChunks[c].size(), (unsigned long long)pairsIn, (unsigned long long)sw);
                }
                std::fprintf(stderr, "cells: HBM-scratch list %zu\n", bigList.size());
            }
            // (A candidate climbs at most CELLS_CLASSES classes and repeats one of them once, in the packed table.)
            for(int round = 0; round < 2 * CELLS_CLASSES + 1; round++) {
                bool any = false;
                if(round == 0 && firstRoundLaunched) any = firstRoundAny;
                else {
                if(members.size() > membersUploaded) {
                    MI355X_ASSERT(members.size() <= memberCapacity);
                    HIP_CHECK(hipMemcpyAsync(b.pairList.data() + membersUploaded, members.data() + membersUploaded,
                        (members.size() - membersUploaded) * 4, hipMemcpyHostToDevice, stream));
                    membersUploaded = members.size();
                }
                // The three classes' chunk lists go up side by side and their launches follow each other without the host
                // in between (the lists are reused once the flags below have come back).
                size_t chunkTotal = 0;
                for(int c = 0; c < CELLS_CLASSES; c++) chunkTotal += classChunks[c].size();
                b.chunks.reserve(chunkTotal, stream);
                size_t chunkOffset = 0;
                for(int c = 0; c < CELLS_CLASSES; c++) {
                    std::vector<CellsChunk>& list = classChunks[c];
                    if(list.empty()) continue;
                    any = true;
                    HIP_CHECK(hipMemcpyAsync(b.chunks.data() + chunkOffset, list.data(), list.size() * sizeof(CellsChunk), hipMemcpyHostToDevice, stream));
                    uint64_t bytes = 0, candidatesIn = 0;
                    for(const CellsChunk& ch : list) {
                        candidatesIn += ch.count;
                        for(uint32_t q = 0; q < ch.count; q++) { const PairDesc& pd = hostPairs[members[ch.firstMember + q]]; bytes += 4ULL * (uint64_t(pd.nx) + pd.ny); }
                    }
                    launchCellsChunks(ctx, ws, b, c, b.chunks.data() + chunkOffset, uint32_t(list.size()), opt, magicX, magicY, taskCapacity, bytes, candidatesIn, hitLists);
                    chunkOffset += list.size();
                }
                }
                if(!any) break;
                // Candidates that overflowed their tables climb one class.  (The task counters come with the flags: if nothing climbs and nothing
                // goes on to the HBM-scratch kernel, they are the batch's final ones -- one synchronisation instead of three.)
                HIP_CHECK(hipMemcpyAsync(hostFlags.data(), b.pairFlags.data(), n, hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipMemcpyAsync(hostCounters, b.counters.data(), sizeof(hostCounters), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
                flagsAndCountersCurrent = true;
                for(int c = 0; c < CELLS_CLASSES; c++) classChunks[c].clear();
                bool retry = false;
                uint64_t reasonHistogram[16] = {0};
                for(uint32_t k = 0; k < n; k++) {
                    if((hostFlags[k] & 0x0f) != PAIR_RESOURCE) continue;
                    if(debug) ++reasonHistogram[hostFlags[k] >> 4];
                    retry = true;
                    // Retry alone in the next class whose table holds one of the two reads (read 0 if both fit) -- or, when a byte
                    // of its cell grid overflowed (bit 7), in the same class once more with its cells in the packed table.
                    const bool gridOverflow = (hostFlags[k] & 0x80) != 0 && !pairNoGrid[k];
                    if(gridOverflow) pairNoGrid[k] = 1;
                    int c = pairClass[k] + (gridOverflow ? 0 : 1);
                    bool sw = false;
                    const int reasons = hostFlags[k] >> 4;            // 1 the cell table, 2 the kept list, 4 geometry (8: the grid byte, handled above)
                    for(; c < CELLS_LONG; c++) {
                        const uint64_t cap = 1ULL << CELLS_NA_LOG2[c];
                        if(hostPairs[k].nx < cap) { sw = false; break; }
                        if(hostPairs[k].ny < cap) { sw = true; break; }
                    }
                    if(c >= CELLS_LONG) {
                        // Beyond the last chunk class, or out of the windowed one: only the kept list too short (more kept cells than a
                        // wavefront's registers hold: a long overlap) has somewhere to go in LDS -- the windowed class's large graph, which
                        // takes any two reads the windowed class takes; a full cell table goes on to the HBM-scratch kernel.
                        const bool keptOnly = (reasons & 2) != 0 && (reasons & 5) == 0;
                        c = (pairClass[k] < CELLS_LONG_BIG && keptOnly && cellsLongGeometryOk(classRule, hostPairs[k].nx, hostPairs[k].ny)) ? CELLS_LONG_BIG : CELLS_CLASSES;
                    }
                    hostFlags[k] = 0;
                    pairClass[k] = c;
                    if(c == CELLS_LONG_BIG) pairSlotsLog2[k] = estimateLog2(k);
                    if(c >= CELLS_CLASSES) { bigList.push_back(k); bigLog2.push_back(estimateLog2(k)); }
                    else addChunk(&k, 1, sw, c, pairNoGrid[k] != 0);
                }
                if(!retry) break;
                flagsAndCountersCurrent = false;
                if(debug) {
                    for(int c = 0; c < CELLS_CLASSES; c++) std::fprintf(stderr, "cells: round %d retries -> class %d: %zu\n", round, c, classChunks[c].size());
                    std::fprintf(stderr, "cells: round %d HBM-scratch list now %zu; reasons cell-table %llu kept-list %llu geometry %llu both %llu grid-byte %llu\n", round, bigList.size(),
                        (unsigned long long)reasonHistogram[1], (unsigned long long)reasonHistogram[2], (unsigned long long)reasonHistogram[4],
                        (unsigned long long)(reasonHistogram[3] + reasonHistogram[5] + reasonHistogram[6] + reasonHistogram[7]),
                        (unsigned long long)(reasonHistogram[8] + reasonHistogram[9] + reasonHistogram[10] + reasonHistogram[11] + reasonHistogram[12] + reasonHistogram[13] + reasonHistogram[14] + reasonHistogram[15]));
                }
                HIP_CHECK(hipMemcpyAsync(b.pairFlags.data(), hostFlags.data(), n, hipMemcpyHostToDevice, stream));
            }
            if(!flagsAndCountersCurrent) {
                HIP_CHECK(hipMemcpyAsync(hostFlags.data(), b.pairFlags.data(), n, hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
            }
            b.pairList.reserve(n, stream);
            const uint64_t scratchWordCap = 1ULL << 31;             // 8 GiB of HBM scratch per launch
            while(!bigList.empty()) {
                flagsAndCountersCurrent = false;
                // Clear the flags of the candidates about to be retried.
                for(uint32_t k : bigList) hostFlags[k] = 0;
                HIP_CHECK(hipMemcpyAsync(b.pairFlags.data(), hostFlags.data(), n, hipMemcpyHostToDevice, stream));
                size_t begin = 0;
                while(begin < bigList.size()) {
                    std::vector<uint64_t> offsets;
                    uint64_t words = 0;
                    size_t end = begin;
                    while(end < bigList.size()) {
                        const uint64_t need = (9ULL << bigLog2[end]) / 2;              // 4.5 words per slot
                        if(end > begin && words + need > scratchWordCap) break;
                        offsets.push_back(words); words += need; ++end;
                    }
                    const uint32_t count = uint32_t(end - begin);
                    b.bigScratch.reserve(words, stream); b.bigOffsets.reserve(count, stream); b.bigLog2.reserve(count, stream);
                    HIP_CHECK(hipMemcpyAsync(b.pairList.data(), bigList.data() + begin, count * 4ULL, hipMemcpyHostToDevice, stream));
                    HIP_CHECK(hipMemcpyAsync(b.bigOffsets.data(), offsets.data(), count * 8ULL, hipMemcpyHostToDevice, stream));
                    HIP_CHECK(hipMemcpyAsync(b.bigLog2.data(), bigLog2.data() + begin, count, hipMemcpyHostToDevice, stream));
                    // (a candidate that climbed here from a chunk kernel keeps no match list: what the attempt that overflowed wrote is not
                    // the sparse path's to read -- its header says 0xffffffff for "cells computed by the HBM kernel")
                    if(listHits) { hipLaunchKernelGGL(hitListNoneKernel, dim3(divUp(count, 256)), dim3(256), 0, stream, (const uint32_t*)b.pairList.data(), count, b.hitMeta.data()); HIP_CHECK(hipGetLastError()); }
                    uint64_t bigBytes = 0;
                    for(size_t q = begin; q < end; q++) bigBytes += 4ULL * (uint64_t(hostPairs[bigList[q]].nx) + hostPairs[bigList[q]].ny);
                    SHASTA_TIMED(ctx, "align4CellsKernel<true, false>", stream, bigBytes, count,
                        hipLaunchKernelGGL(align4CellsKernel<true>, dim3(count), dim3(CELLS_THREADS), 0, stream,
                            (const uint32_t*)ctx.kmerIds.data(), (const PairDesc*)b.pairs.data(), (const uint32_t*)b.pairList.data(), count, opt,
                            b.tasks.data(), b.counters.data(), taskCapacity, b.pairFlags.data(),
                            b.bigScratch.data(), (const uint64_t*)b.bigOffsets.data(), (const uint8_t*)b.bigLog2.data(),
                            (uint32_t*)nullptr, (const uint64_t*)nullptr, (uint32_t*)nullptr));
                    for(size_t q = begin; q < end; q++) pairSlotsLog2[bigList[q]] = bigLog2[q];
                    HIP_CHECK(hipGetLastError());
                    HIP_CHECK(hipStreamSynchronize(stream));
                    begin = end;
                }
                HIP_CHECK(hipMemcpyAsync(hostFlags.data(), b.pairFlags.data(), n, hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
                std::vector<uint32_t> nextList; std::vector<uint8_t> nextLog2;
                for(size_t q = 0; q < bigList.size(); q++) {
                    if(hostFlags[bigList[q]] == PAIR_RESOURCE && bigLog2[q] < 24) {
                        nextList.push_back(bigList[q]); nextLog2.push_back(uint8_t(std::min(24, bigLog2[q] + 3)));
                    }
                }
                bigList.swap(nextList); bigLog2.swap(nextLog2);
            }
        }
        {   // (one copy, one synchronisation for both counters -- unless they came with the flags)
            if(!flagsAndCountersCurrent) {
                HIP_CHECK(hipMemcpyAsync(hostCounters, b.counters.data(), sizeof(hostCounters), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
            }
            taskCount = hostCounters[0];
            wideCount = hostCounters[CELLS_WIDE_COUNTER];      // (components / step-2 bands of more than 1024 diagonals, listed from the back)
        }
        if(uint64_t(taskCount) + wideCount <= taskCapacity) break;
        // More DP tasks than the list was sized for (many small components per candidate: low-complexity
        // reads, or options that keep nearly every cell).  The count is exact -- stores past the
        // capacity were dropped, the counter was not -- so the stage runs once more with room for all.
        if(m3) throw std::runtime_error("Align3: task list overflow.");
        if(std::getenv("SHASTA_MI355X_DEBUG")) std::fprintf(stderr, "cells: %u DP tasks exceed the capacity %u: running the stage again\n", taskCount, taskCapacity);
        taskCapacity = taskCount + wideCount + 1024;
        b.tasks.reserve(taskCapacity, stream);
        HIP_CHECK(hipMemsetAsync(b.counters.data(), 0, 16 * sizeof(uint32_t), stream));
        HIP_CHECK(hipMemsetAsync(b.pairFlags.data(), 0, n, stream));
        }
        // The wide components come forward to tasks[taskCount ...) (their results and aligned pairs follow the others').
        std::vector<DpTask> wideTasksHost(wideCount);
        if(wideCount) {
            HIP_CHECK(hipMemcpyAsync(wideTasksHost.data(), b.tasks.data() + (taskCapacity - wideCount), wideCount * sizeof(DpTask), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            // (the order the atomics gave them is the order of no-one's choosing: by candidate and component, so that runs repeat)
            std::sort(wideTasksHost.begin(), wideTasksHost.end(), [](const DpTask& x, const DpTask& y) { return x.pair != y.pair ? x.pair < y.pair : x.label < y.label; });
            HIP_CHECK(hipMemcpyAsync(b.tasks.data() + taskCount, wideTasksHost.data(), wideCount * sizeof(DpTask), hipMemcpyHostToDevice, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
        }

        phaseCells = phaseMs(phaseStart);
        // K10: sort the tasks by (band class, length), bundle, forward DP, traceback.
        if(taskCount + wideCount) {
            const SparseInput sparseInput{b.hits.data(), b.hitBase.data(), b.hitMeta.data(), maxOrdered};
            b.pairsWanted = wantOrdinals;
            // Align method 3 keeps the classes of its second step on the worker's main stream (no events: no fork) unless
            // SHASTA_MI355X_DP_FORK=1 asks for the fork: the one difference between two runs this library has shown was a method-3
            // call, every method-3 task runs the dense kernels, and those kernels on ONE stream are what 64 million task
            // executions on the device showed to be repeatable (profiles/r06_flake5_dp_unit_40000.log) -- the fork's own A/B
            // (scripts/gpu_r06_flake8.sh) has not run.  Method 4 forks as before: its dense remainder is what its measurements had.
            const bool forkMethod3 = [] { const char* e = std::getenv("SHASTA_MI355X_DP_FORK"); return e && e[0] == '1'; }();
            out.dpCells += runDpTasks(ctx, ws, b, taskCount, dpOpt, (m3 && !forkMethod3) ? nullptr : &w.ev, &out.dpStats, m3 ? &m3->scores : nullptr, &wideTasksHost, &hostPairs, listHits ? &sparseInput : nullptr);
            out.hadTasks = true;
            const uint32_t allTasks = taskCount + wideCount;
            SHASTA_TIMED(ctx, "winnerKernel", stream, 0, allTasks,
                hipLaunchKernelGGL(winnerKernel, dim3(divUp(allTasks, 256)), dim3(256), 0, stream,
                    (const DpTask*)b.tasks.data(), (const DpResult*)b.results.data(), allTasks,
                    (const unsigned long long*)b.pairBest.data(), b.pairWinner.data(), b.pairTie.data(), b.counters.data() + 12));
            HIP_CHECK(hipGetLastError());
            // Candidates whose best components tie on markerCount: the reference's component order decides (method 4 only:
            // method 3 has one alignment per candidate).  Whether there is one comes with the batch's totals below (round 6: it was
            // a synchronisation of its own between the DP and the filters); the rare batch that has one resolves it then and runs
            // the filters again.
            tieCounterWanted = !m3;
            tieTasks = allTasks;
        } else {
            b.results.reserve(1, stream); b.ordScratch.reserve(2, stream);
            b.sparseStateTasks = 0; b.streamsInLists = false;
        }

        if(debugPhases) { HIP_CHECK(hipStreamSynchronize(stream)); phaseDp = phaseMs(phaseStart) - phaseCells; }
        // K11.
        const unsigned gp = divUp(uint64_t(n) + 1, 256);
        const unsigned gw = divUp((uint64_t(n) + 1) * WAVE, 256);
        uint32_t storedCount = 0;
        uint64_t ordTotalOut = 0, byteTotal = 0;
        for(int attempt = 0; ; attempt++) {
        const KernelTimers::Span finalizeSpan = ctx.timers.begin("finalizeKernel + scans", stream);
        hipLaunchKernelGGL(finalizeKernel, dim3(gp), dim3(256), 0, stream,
            (const PairDesc*)b.pairs.data(), (const shasta_oriented_read_pair*)b.candidates.data(), n,
            (const DpResult*)b.results.data(), (const unsigned long long*)b.pairBest.data(),
            (const uint32_t*)b.pairWinner.data(), (const uint8_t*)b.pairTie.data(), (const uint8_t*)b.pairFlags.data(),
            opt, wantOrdinals ? 1 : 0, b.status.data(), b.rows.data(), b.storedFlags.data(), b.ordCounts.data(), b.sizes.data());
        exclusiveScan<uint32_t>(b.storedFlags.data(), b.storedIndex.data(), uint64_t(n) + 1, b.scanTemp32.data(), stream);
        b.scanTemp64.reserve(scanTempElements(uint64_t(n) + 1), stream);
        exclusiveScan<uint64_t>(b.ordCounts.data(), b.ordCounts.data(), uint64_t(n) + 1, b.scanTemp64.data(), stream);
        // (the sizes of the compressed alignments were counted by dpMetricsKernel in its pass over the pairs)
        exclusiveScan<uint64_t>(b.sizes.data(), b.sizes.data(), uint64_t(n) + 1, b.scanTemp64.data(), stream);
        (void)ctx.timers.end(finalizeSpan, 64ULL * n, n);
        HIP_CHECK(hipGetLastError());
        // The three totals and the tie counter into page-locked memory, one synchronisation for all four.
        uint64_t* const totals = static_cast<uint64_t*>(b.pinTotals.reserve(4 * sizeof(uint64_t)));
        totals[0] = totals[1] = totals[2] = totals[3] = 0;
        HIP_CHECK(hipMemcpyAsync(&totals[0], b.storedIndex.data() + n, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipMemcpyAsync(&totals[1], b.ordCounts.data() + n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipMemcpyAsync(&totals[2], b.sizes.data() + n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        if(tieCounterWanted) HIP_CHECK(hipMemcpyAsync(&totals[3], b.counters.data() + 12, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        if(tieCounterWanted && uint32_t(totals[3]) != 0 && attempt == 0) {
            resolveComponentTies(ctx, ws, b, n, tieTasks, hostPairs, pairClass, pairSlotsLog2, pairNoGrid, opt, cellsMagicX, cellsMagicY);
            continue;                                     // (the filters and the scans once more, on the winners as they are now)
        }
        storedCount = uint32_t(totals[0]); ordTotalOut = totals[1]; byteTotal = totals[2];
        break;
        }
        uint8_t* const bytesPlace = publishSizes(batchIndex, storedCount, byteTotal, wantOrdinals ? ordTotalOut : 0);
        b.bytes.reserve(byteTotal + 1, stream);
        const KernelTimers::Span writeSpan = ctx.timers.begin("compressWriteKernel", stream);
        hipLaunchKernelGGL(compressWriteKernel, dim3(gw), dim3(256), 0, stream,
            (const uint32_t*)b.storedFlags.data(), (const uint32_t*)b.storedIndex.data(), (const DpResult*)b.results.data(),
            (const uint32_t*)b.pairWinner.data(), (const uint32_t*)b.ordScratch.data(), n,
            (const uint64_t*)b.sizes.data(), b.bytes.data(), b.compressedToc.data(),
            (const shasta_alignment_data*)b.rows.data(), b.rowsOut.data(),
            (b.sparseStateTasks && b.streamsInLists) ? (const uint8_t*)b.sparseState.data() : (const uint8_t*)nullptr, b.sparseStateTasks,
            (const PairDesc*)b.pairs.data(), (const uint64_t*)b.ordCap.data(), (const uint32_t*)b.sparseSorted.data());
        const size_t writeHandle = ctx.timers.end(writeSpan, 0, storedCount);
        HIP_CHECK(hipGetLastError());

        // Copy this batch's results out (assembled in candidate order once every batch is done).
        // Device -> pinned staging (asynchronous, PCIe speed) -> the batch's output vectors.
        void* pinRows = b.pinRows.reserve(storedCount * sizeof(shasta_alignment_data));
        void* pinToc = b.pinToc.reserve(storedCount * sizeof(uint64_t));
        void* pinBytes = bytesPlace ? static_cast<void*>(bytesPlace) : b.pinBytes.reserve(byteTotal);      // (its place in the page-locked array, or the staging buffer)
        void* pinStatus = b.pinStatus.reserve(n);
        void* pinOrdToc = nullptr; void* pinOrdinals = nullptr;
        if(storedCount) {
            HIP_CHECK(hipMemcpyAsync(pinRows, b.rowsOut.data(), storedCount * sizeof(shasta_alignment_data), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipMemcpyAsync(pinToc, b.compressedToc.data(), storedCount * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        }
        if(byteTotal) HIP_CHECK(hipMemcpyAsync(pinBytes, b.bytes.data(), byteTotal, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipMemcpyAsync(pinStatus, b.status.data(), n, hipMemcpyDeviceToHost, stream));
        if(wantOrdinals) {
            pinOrdToc = b.pinOrdToc.reserve((uint64_t(n) + 1) * sizeof(uint64_t));
            pinOrdinals = b.pinOrdinals.reserve(2 * ordTotalOut * sizeof(uint32_t));
            HIP_CHECK(hipMemcpyAsync(pinOrdToc, b.ordCounts.data(), (uint64_t(n) + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
            if(ordTotalOut) {
                b.ordOut.reserve(2 * ordTotalOut, stream);
                hipLaunchKernelGGL(gatherOrdinalsKernel, dim3(divUp(uint64_t(n) * 64, 256)), dim3(256), 0, stream,
                    (const DpResult*)b.results.data(), (const uint32_t*)b.pairWinner.data(), (const uint64_t*)b.ordCounts.data(), n,
                    (const uint32_t*)b.ordScratch.data(), b.ordOut.data());
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipMemcpyAsync(pinOrdinals, b.ordOut.data(), 2 * ordTotalOut * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            }
        }
        HIP_CHECK(hipStreamSynchronize(stream));
        out.rows.assign(static_cast<const shasta_alignment_data*>(pinRows), static_cast<const shasta_alignment_data*>(pinRows) + storedCount);
        hostToc64.assign(static_cast<const uint64_t*>(pinToc), static_cast<const uint64_t*>(pinToc) + storedCount);
        out.bytesInPlace = bytesPlace != nullptr;
        out.stagedBytes = bytesPlace ? nullptr : static_cast<const uint8_t*>(pinBytes); out.byteCount = byteTotal;      // (placeFinished takes them from there)
        std::memcpy(outStatus.data() + batchBegin, pinStatus, n);
        if(wantOrdinals) {
            out.ordToc.assign(static_cast<const uint64_t*>(pinOrdToc), static_cast<const uint64_t*>(pinOrdToc) + uint64_t(n) + 1);
            out.ordinals.assign(static_cast<const uint32_t*>(pinOrdinals), static_cast<const uint32_t*>(pinOrdinals) + 2 * ordTotalOut);
        }
        // CSR of CompressedAlignments: end offset of each stored alignment, relative to this batch.
        out.tocEnds.resize(storedCount);
        for(uint32_t k = 0; k < storedCount; k++) out.tocEnds[k] = (k + 1 < storedCount ? hostToc64[k + 1] : byteTotal);
        for(uint32_t k = 0; k < storedCount; k++) out.alignedBytes += 8ULL * out.rows[k].info.markerCount;
        // Both compress kernels read the 8-byte ordinal pairs of the stored alignments; the second writes the blobs and the 64-byte rows.
        // (round 5: the wave kernel leaves its alignments in shasta::compress form, which this kernel copies -- the bytes are read and written,
        // the rows read and written; only the other tasks' alignments, a tenth, are still made from their aligned pairs: booked as copies all)
        ctx.timers.amend(writeHandle, (b.streamsInLists ? byteTotal : out.alignedBytes) + byteTotal + 128ULL * storedCount, storedCount);
        if(debugPhases) {
            phaseFinish = phaseMs(phaseStart) - phaseCells - phaseDp;
            std::fprintf(stderr, "batch %llu (%u candidates, %u tasks): candidates -> DP tasks %.1f ms, DP %.1f ms, filters + compression + copies %.1f ms\n",
                (unsigned long long)batchIndex, n, taskCount, phaseCells, phaseDp, phaseFinish);
        }
    };

    // Borrowed results live in arrays the context keeps from call to call: a batch is copied to its place in them at once, by
    // the worker that finished it, while the other workers' batches are still on the device -- from the worker's staging buffer
    // to its place, one host copy.  Its place is the sum of what the batches before it put out, known as soon as they have
    // published their sizes (publishSizes: before their results are even written), so the order in which batches FINISH does
    // not matter (the short last batch of a call usually finishes before its predecessor).  A batch that does not fit the
    // arrays as they are (first call, or more output than last time), or whose predecessors have not published yet, keeps its
    // bytes in a vector of its own and waits for the end, where the arrays grow.
    struct Placement { uint64_t rowBase = 0, byteBase = 0, ordBase = 0, rows = 0, bytes = 0, ords = 0; bool reserved = false, fits = false, placed = false; };
    std::vector<Placement> placements(batchCount);
    std::vector<char> sized(batchCount, 0);
    std::mutex placeMutex;
    uint64_t nextToPlace = 0, placeRows = 0, placeBytes = 0, placeOrdinals = 0;
    bool placeEarly = borrowed;
    if(placeEarly) {
        // Rows and offsets have a bound (one per candidate); bytes and ordinals take what the arrays already hold.
        if(store.rows.size() < candidateCount) store.rows.resize(std::max<uint64_t>(1, candidateCount));
        if(store.compressedToc.size() < candidateCount + 1) store.compressedToc.resize(candidateCount + 1);
        if(wantOrdinals && store.ordinalsToc.size() < candidateCount + 1) store.ordinalsToc.resize(candidateCount + 1);
    }
    publishSizes = [&](uint64_t k, uint64_t rows, uint64_t bytes, uint64_t ords) -> uint8_t* {
        std::lock_guard<std::mutex> lock(placeMutex);
        placements[k].rows = rows; placements[k].bytes = bytes; placements[k].ords = ords;
        sized[k] = 1;
        while(nextToPlace < batchCount && sized[nextToPlace]) {
            Placement& at = placements[nextToPlace];
            at.rowBase = placeRows; at.byteBase = placeBytes; at.ordBase = placeOrdinals;
            placeRows += at.rows; placeBytes += at.bytes; placeOrdinals += at.ords;
            if(placeEarly && (placeBytes > store.bytes.size() || (wantOrdinals && 2 * placeOrdinals > store.ordinals.size()))) placeEarly = false;
            at.fits = placeEarly; at.reserved = true;
            ++nextToPlace;
        }
        const Placement& mine = placements[k];
        return (mine.reserved && mine.fits && mine.bytes && store.bytes.pageLocked()) ? store.bytes.data() + mine.byteBase : nullptr;
    };
    // (SHASTA_MI355X_SLICE_COPY_MIN_BYTES: from how many bytes on the tail's copy is cut into slices, default 8 MiB -- tests set 1.)
    static const uint64_t sliceCopyMinimum = [] { const char* e = std::getenv("SHASTA_MI355X_SLICE_COPY_MIN_BYTES"); return e ? std::max<uint64_t>(1, std::strtoull(e, nullptr, 10)) : (8ULL << 20); }();
    // (`threads` > 1: the copy of the call's last batches, when the other workers have nothing left to do and the device waits
    // for the host: the bytes in as many slices.)
    auto copyBatch = [&](uint64_t k, const Placement& at, shasta_alignment_data* rows, uint64_t* toc, uint8_t* bytes, uint64_t* ordinalsToc, uint32_t* ordinals, int threads = 1) {
        const BatchOutput& o = outputs[k];
        if(!o.rows.empty()) std::memcpy(rows + at.rowBase, o.rows.data(), o.rows.size() * sizeof(shasta_alignment_data));
        if(o.byteCount && !o.bytesInPlace) {
            const uint8_t* from = o.stagedBytes ? o.stagedBytes : o.bytes.data();
            uint8_t* to = bytes + at.byteBase;
            if(threads > 1 && o.byteCount >= sliceCopyMinimum) {
                const uint64_t slice = (o.byteCount / uint64_t(threads) + 4095) & ~4095ULL;
                std::vector<std::thread> helpers;
                for(uint64_t begin = slice; begin < o.byteCount; begin += slice) {
                    const uint64_t count = std::min<uint64_t>(slice, o.byteCount - begin);
                    helpers.emplace_back([to, from, begin, count] { std::memcpy(to + begin, from + begin, count); });
                }
                std::memcpy(to, from, std::min<uint64_t>(slice, o.byteCount));
                for(std::thread& t : helpers) t.join();
            } else {
                std::memcpy(to, from, o.byteCount);
            }
        }
        for(size_t q = 0; q < o.tocEnds.size(); q++) toc[at.rowBase + q + 1] = at.byteBase + o.tocEnds[q];
        if(wantOrdinals) {
            if(!o.ordinals.empty()) std::memcpy(ordinals + 2 * at.ordBase, o.ordinals.data(), o.ordinals.size() * sizeof(uint32_t));
            for(size_t q = 1; q < o.ordToc.size(); q++) ordinalsToc[batchStart[k] + q] = at.ordBase + o.ordToc[q];
        }
    };
    std::atomic<uint64_t> nextBatch(0);
    // The bytes of a batch out of the worker's staging buffer (which its next batch overwrites) into the batch's own vector.
    auto keepBytes = [&](BatchOutput& o) {
        if(!o.stagedBytes) return;
        o.bytes.assign(o.stagedBytes, o.stagedBytes + o.byteCount);
        o.stagedBytes = nullptr;
    };
    auto placeFinished = [&](uint64_t finished, const shasta_alignment_data* workerRows, hipStream_t workerStream) {
        BatchOutput& own = outputs[finished];
        Placement at;
        bool direct = false, tail = false;
        {
            std::lock_guard<std::mutex> lock(placeMutex);
            Placement& mine = placements[finished];
            MI355X_ASSERT(mine.rows == own.rows.size() && mine.bytes == own.byteCount && mine.ords == own.ordinals.size() / 2);
            direct = mine.reserved && mine.fits;
            MI355X_ASSERT(direct || !own.bytesInPlace);
            if(direct) mine.placed = true;
            at = mine;
            tail = nextBatch.load() >= batchCount;      // every batch has been taken: this worker has none to go on with
        }
        if(direct) {
            // (the table's keys of this batch's rows, while the rows are still in the worker's device buffer: tables.hip)
            if(tableKeys) alignmentTableKeysOfBatch(ctx, workerRows, at.rows, at.rowBase, workerStream);
            copyBatch(finished, at, store.rows.data(), store.compressedToc.data(), store.bytes.data(), store.ordinalsToc.data(), store.ordinals.data(), tail ? 4 : 1);
            own.stagedBytes = nullptr;       // (copied from the staging buffer to its place: nothing else reads it)
        } else {
            keepBytes(own);                  // (the end of the call copies it from its vector)
        }
    };
    auto workerLoop = [&](int k) {
        try {
            HIP_CHECK(hipSetDevice(ctx.device));
            for(;;) {
                const uint64_t batchIndex = nextBatch.fetch_add(1);
                if(batchIndex >= batchCount) break;
                processBatch(workers[k], batchIndex);
                placeFinished(batchIndex, workers[k].scratch->rowsOut.data(), workers[k].stream);
            }
        } catch(const std::exception& e) {
            workers[k].error = e.what();
            nextBatch.store(batchCount);
        }
    };
    {
        std::vector<std::thread> others;
        for(int k = 1; k < workerCount; k++) others.emplace_back(workerLoop, k);
        workerLoop(0);
        for(std::thread& t : others) t.join();
    }
    for(int k = 1; k < workerCount; k++) {
        HIP_CHECK(hipEventRecord(evOther, workers[k].stream));
        HIP_CHECK(hipStreamWaitEvent(ctx.stream, evOther, 0));
    }
    const double msWorkersDone = hostMs();
    HIP_CHECK(hipEventRecord(evEnd, ctx.stream));
    HIP_CHECK(hipStreamSynchronize(ctx.stream));
    const double msDeviceDone = hostMs();
    // Every worker's buffers up to what any worker's batches needed, now that nothing is in flight: which worker meets which
    // batch changes from call to call, and a worker that found a mark above its buffer at the start of a batch of the NEXT call
    // reallocated there -- hipFree waits for the whole device, in the middle of that call (the second call on a context: 192 ms
    // against 146 - 157 for the later ones at 100 k reads).
    for(int k = 0; k < workerCount; k++) if(workers[k].error.empty()) workers[k].scratch->raiseToMarks(workers[k].stream);
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, evBegin, evEnd));
    result.deviceSeconds = ms * 1e-3;
    // (the events are the context's: nothing to destroy)
    for(Worker& w : workers) if(!w.error.empty()) throw std::runtime_error(w.error);

#ifdef SHASTA_PROFILE_PHASES
    {
        unsigned long long h[16];
        HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phaseCycles), sizeof(h)));
        std::fprintf(stderr, "phase cycles (lane-0 sums):");
        for(int k = 0; k < 16; k++) std::fprintf(stderr, " %llu", h[k]);
        std::fprintf(stderr, "\n");
        std::memset(h, 0, sizeof(h));
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phaseCycles), h, sizeof(h)));
    }
#endif

    // Assemble the outputs in candidate order.
    uint64_t rowTotal = 0, byteTotalAll = 0, ordTotalAll = 0, dpCellsTotal = 0, kmerIdBytes = 0, alignedBytes = 0;
    for(uint64_t k = 0; k < batchCount; k++) {
        const BatchOutput& o = outputs[k];
        rowTotal += o.rows.size(); byteTotalAll += o.byteCount; ordTotalAll += o.ordinals.size() / 2;
        dpCellsTotal += o.dpCells; kmerIdBytes += o.kmerIdBytes; alignedBytes += o.alignedBytes;
    }
    if(borrowed) {
        // (grow only: the arrays keep their size from call to call so that batches can be placed early next time)
        if(store.rows.size() < std::max<uint64_t>(1, rowTotal)) store.rows.resize(std::max<uint64_t>(1, rowTotal));
        if(store.compressedToc.size() < rowTotal + 1) store.compressedToc.resize(rowTotal + 1);
        if(store.bytes.size() < std::max<uint64_t>(1, byteTotalAll)) store.bytes.resize(std::max<uint64_t>(1, byteTotalAll) + byteTotalAll / 16);
        result.alignmentData = store.rows.data(); result.compressedToc = store.compressedToc.data();
        result.compressedData = store.bytes.data(); result.status = store.status.data();
        if(wantOrdinals) {
            if(store.ordinalsToc.size() < candidateCount + 1) store.ordinalsToc.resize(candidateCount + 1);
            if(store.ordinals.size() < std::max<uint64_t>(1, 2 * ordTotalAll)) store.ordinals.resize(std::max<uint64_t>(1, 2 * ordTotalAll) + ordTotalAll / 8);
            result.ordinalsToc = store.ordinalsToc.data(); result.ordinals = store.ordinals.data();
            result.ordinalsToc[0] = 0;
        }
        result.owner = &ctx;
    } else {
        auto allocate = [](size_t bytes) { void* p = std::malloc(std::max<size_t>(1, bytes)); if(!p) throw std::bad_alloc(); return p; };
        result.alignmentData = static_cast<shasta_alignment_data*>(allocate(rowTotal * sizeof(shasta_alignment_data)));
        result.compressedToc = static_cast<uint64_t*>(allocate((rowTotal + 1) * sizeof(uint64_t)));
        result.compressedData = static_cast<uint8_t*>(allocate(byteTotalAll));
        result.status = static_cast<uint8_t*>(allocate(candidateCount));
        if(candidateCount) std::memcpy(result.status, outStatus.data(), candidateCount);
        if(wantOrdinals) {
            result.ordinalsToc = static_cast<uint64_t*>(allocate((candidateCount + 1) * sizeof(uint64_t)));
            result.ordinals = static_cast<uint32_t*>(allocate(2 * ordTotalAll * sizeof(uint32_t)));
            result.ordinalsToc[0] = 0;
        }
    }
    result.compressedToc[0] = 0;
    // What was not placed early: every batch's share of the output arrays is known now, the copies (a few hundred megabytes when
    // it is all of them) run on as many host threads as there were workers instead of one.
    MI355X_ASSERT(nextToPlace == batchCount && placeRows == rowTotal && placeBytes == byteTotalAll);
    auto assemble = [&](uint64_t first, uint64_t stride) {
        for(uint64_t k = first; k < batchCount; k += stride) {
            // (one or two batches left: each copy in slices on several threads instead of the batches side by side)
            if(!placements[k].placed) copyBatch(k, placements[k], result.alignmentData, result.compressedToc, result.compressedData, result.ordinalsToc, result.ordinals, stride <= 2 ? 4 : 1);
        }
    };
    {
        uint64_t left = 0;
        for(const Placement& at : placements) left += at.placed ? 0 : 1;
        const uint64_t threads = std::min<uint64_t>(uint64_t(workerCount), std::max<uint64_t>(1, left));
        std::vector<std::thread> others;
        for(uint64_t k = 1; k < threads; k++) others.emplace_back(assemble, k, threads);
        assemble(0, threads);
        for(std::thread& t : others) t.join();
    }

    if(borrowed) { store.lastRowCount = rowTotal; store.valid = true; }
    result.alignmentCount = rowTotal;
    result.dpCellCount = dpCellsTotal;
    result.kmerIdBytes = kmerIdBytes;
    result.alignedBytes = alignedBytes;
    result.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if(logHost) std::fprintf(stderr, "alignRun host clock: first event at %.2f ms, workers joined at %.2f, device done at %.2f, return at %.2f (%llu batches)\n",
        msAtBegin, msWorkersDone, msDeviceDone, result.seconds * 1e3, (unsigned long long)batchCount);
}

}  // namespace

// The AlignmentData rows the context's last borrowed aligner call stored (tables.hip builds the alignment table from them).
const shasta_alignment_data* borrowedAlignmentRows(Context& ctx, uint64_t* count)
{
    AlignStore* store = static_cast<AlignStore*>(ctx.alignStore.get());
    if(!store || !store->valid) throw std::runtime_error("alignment_table: the context holds no alignments (it follows a *_run_borrowed call).");
    *count = store->lastRowCount;
    return store->rows.data();
}

void align4Run(Context& ctx, uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align4_options& options, bool wantOrdinals, shasta_align4_result& result, bool borrowed)
{
    alignRun(ctx, candidateCount, candidates, makeOptions(options), nullptr, wantOrdinals, result, borrowed);
}

// Align method 3 on the resident markers (Assembler::alignOrientedReads3 for every candidate, then
// the filters of src/AssemblerAlign.cpp:439-472).  Same result layout as align4Run.
void align3Run(Context& ctx, uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align3_options& o, bool wantOrdinals, shasta_align4_result& result, bool borrowed)
{
    // Scores as the reference passes them to SeqAn (src/AssemblerAlign3.cpp:22-33, 120, 257): any triple whose sums cannot
    // leave the DP kernels' 32-bit range -- |score| (nx + ny) stays below 2^28 for the longest pair the kernels take (2^25
    // markers) when the magnitudes add up to less than 8; larger ones are checked against the call's longest pair below.
    const int64_t scoreMagnitude = std::llabs(o.matchScore) + std::llabs(o.mismatchScore) + std::llabs(o.gapScore);
    if(scoreMagnitude >= (1LL << 20)) throw std::runtime_error("Align3: scores of magnitude 2^20 or more are not supported.");
    if(o.k < 1 || o.k > 16) throw std::runtime_error("Align3: k must be in [1, 16].");
    if(!(o.downsamplingFactor >= 0. && o.downsamplingFactor <= 1.)) throw std::runtime_error("Align3: downsamplingFactor must be in [0, 1].");
    if(o.bandExtend < 0 || o.bandExtend > (1 << 20)) throw std::runtime_error("Align3: bandExtend must be in [0, 2^20].");
    // (A step-2 band of more than 1024 diagonals runs in the wide DP over its band, as Align4's wide components do.)
    if(o.maxBand < 0 || o.maxBand >= ALIGN3_HUGE_MAX_DIAGONALS) throw std::runtime_error("Align3: maxBand must be in [0, 65535] (the widest band the wide DP holds).");
    Align3Plan plan;
    plan.k = uint32_t(o.k);
    plan.hashThreshold = uint32_t(o.downsamplingFactor * double(std::numeric_limits<uint32_t>::max()));   // src/AssemblerAlign3.cpp:71-72
    plan.bandExtend = int32_t(o.bandExtend); plan.maxBand = int32_t(o.maxBand);
    plan.scores = DpScores{int32_t(o.matchScore), int32_t(o.mismatchScore), int32_t(o.gapScore)};
    DeviceOptions opt;
    std::memset(&opt, 0, sizeof(opt));
    opt.deltaX = opt.deltaY = 1;
    opt.minAlignedMarkerCount = o.minAlignedMarkerCount;
    opt.minAlignedFraction = o.minAlignedFraction;
    opt.maxSkip = o.maxSkip; opt.maxDrift = o.maxDrift; opt.maxTrim = o.maxTrim; opt.maxBand = uint64_t(o.maxBand);
    opt.suppressContainments = o.suppressContainments ? 1u : 0u;
    alignRun(ctx, candidateCount, candidates, opt, &plan, wantOrdinals, result, borrowed);
}

void align4Free(shasta_align4_result& r)
{
    if(r.owner) { std::memset(&r, 0, sizeof(r)); return; }     // borrowed: the arrays belong to the context
    std::free(r.alignmentData); std::free(r.compressedToc); std::free(r.compressedData);
    std::free(r.status); std::free(r.ordinalsToc); std::free(r.ordinals);
    std::memset(&r, 0, sizeof(r));
}

// Unit seam: K10 on many (pair, band) tasks at once -- sorted, bundled and run exactly as the tasks of an Align4 batch
// are, so that wavefronts hold several tasks of different geometry.  Task t aligns kmerIds[begin0[t] .. +nx[t]) with
// kmerIds[begin1[t] .. +ny[t]) inside [bandMin[t], bandMax[t]].  Results in task order: counts[t] aligned pairs,
// scores[t], and the pairs themselves concatenated in ordinals.
void bandedDpManyUnit(const uint32_t* kmerIds, uint64_t kmerCount, uint64_t taskCount,
    const uint64_t* begin0, const uint32_t* nx, const uint64_t* begin1, const uint32_t* ny, const int32_t* bandMin, const int32_t* bandMax,
    uint64_t* counts, int32_t* scores, uint32_t* ordinals, uint64_t capacity, double* seconds, uint64_t* cells)
{
    int device = 0;
    HIP_CHECK(hipGetDevice(&device));
    Context ctx(device);
    if(seconds) for(int c = 0; c <= DP_CLASSES; c++) seconds[c] = 0.;
    if(cells) for(int c = 0; c < DP_CLASSES; c++) cells[c] = 0;
    if(taskCount == 0) return;
    if(taskCount > (1u << 24)) throw std::runtime_error("banded_dp_many: too many tasks.");
    std::vector<PairDesc> pairs(taskCount);
    std::vector<DpTask> tasks(taskCount);
    for(uint64_t t = 0; t < taskCount; t++) {
        if(nx[t] == 0 || ny[t] == 0 || begin0[t] + nx[t] > kmerCount || begin1[t] + ny[t] > kmerCount) throw std::runtime_error("banded_dp_many: a sequence is empty or outside kmerIds.");
        if(uint64_t(nx[t]) + uint64_t(ny[t]) >= (1ULL << 25) - 4) throw std::runtime_error("banded_dp_many: two sequences of 2^25 - 4 elements or more between them.");
        if(bandMin[t] > bandMax[t] || int64_t(bandMax[t]) - bandMin[t] + 1 > int64_t(ALIGN3_HUGE_MAX_DIAGONALS)) throw std::runtime_error("banded_dp_many: band width must be in [1, 65536].");
        if(bandMin[t] > int32_t(nx[t]) || bandMax[t] < -int32_t(ny[t])) throw std::runtime_error("banded_dp_many: the band misses the matrix.");
        pairs[t].begin0 = begin0[t]; pairs[t].begin1 = begin1[t]; pairs[t].nx = nx[t]; pairs[t].ny = ny[t];
    }
    // The tasks of up to 1024 diagonals first (the banded DP kernels), the wider ones behind them (the wide DP), as alignRun
    // leaves them; position[k] = the input task of list entry k.
    std::vector<uint32_t> position;
    std::vector<DpTask> wideTasks;
    for(int wide = 0; wide < 2; wide++) {
        for(uint64_t t = 0; t < taskCount; t++) {
            if((bandMax[t] - bandMin[t] + 1 > 1024) != (wide == 1)) continue;
            DpTask task; task.pair = uint32_t(t); task.bandMin = bandMin[t]; task.bandMax = bandMax[t]; task.label = 0;
            tasks[position.size()] = task;
            if(wide) wideTasks.push_back(task);
            position.push_back(uint32_t(t));
        }
    }
    const uint32_t narrowCount = uint32_t(taskCount - wideTasks.size());
    std::vector<uint64_t> toc = {0, kmerCount / 2, kmerCount};
    ctx.setMarkers(1, toc.data(), nullptr, kmerIds, nullptr);
    hipStream_t stream = ctx.stream;
    BatchScratch b;
    b.pairs.reserve(taskCount, stream); b.tasks.reserve(taskCount, stream); b.layoutZeroed(taskCount, stream);
    HIP_CHECK(hipMemcpyAsync(b.pairs.data(), pairs.data(), taskCount * sizeof(PairDesc), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(b.tasks.data(), tasks.data(), taskCount * sizeof(DpTask), hipMemcpyHostToDevice, stream));

    DeviceOptions opt;
    std::memset(&opt, 0, sizeof(opt));
    opt.deltaX = 200; opt.deltaY = 10; opt.maxSkip = opt.maxDrift = opt.maxTrim = ~0ULL; opt.maxBand = 1024;
    // (with the side stream a worker of a batch has: the wide-band classes fork to it and join again, as they do in a call)
    if(!ctx.wideStream[0]) HIP_CHECK(hipStreamCreateWithFlags(&ctx.wideStream[0], hipStreamNonBlocking));
    const WorkStream ws{ctx.stream, &ctx.sortWs, ctx.wideStream[0]};
    DpEvents ev;
    DpBatchStats stats;
    // The sparse path of an Align4 batch (align4_sparse.hpp) takes its matches from the cells kernel; here the host lists them --
    // every (x, y) with equal kmer ids, whatever the band, as the cells kernel would, in an order of no meaning -- so that the seam
    // runs the tasks the way a batch does: sparse where the optimal chain is unique, dense otherwise.  Odd tasks say that read 0
    // was the streamed one (either read may be).
    SparseInput sparseInput{nullptr, nullptr, nullptr, 0};
    const bool sparse = sparseDpEnabled();
    if(sparse) {
        std::vector<uint64_t> hitBase(taskCount + 1, 0);
        std::vector<uint32_t> meta(taskCount), hostHits;
        for(uint64_t t = 0; t < taskCount; t++) hitBase[t + 1] = hitBase[t] + ((nx[t] < 65535 && ny[t] < 65535) ? hitListCapacity(nx[t], ny[t], 13) : 0u);
        hostHits.resize(hitBase[taskCount] + 1);
        std::unordered_map<uint32_t, std::vector<uint32_t>> where;
        for(uint64_t t = 0; t < taskCount; t++) {
            const uint32_t capacity = uint32_t(hitBase[t + 1] - hitBase[t]);
            if(capacity == 0) { meta[t] = HIT_LIST_NONE; continue; }
            where.clear();
            for(uint32_t y = 0; y < ny[t]; y++) where[kmerIds[begin1[t] + y]].push_back(y);
            uint64_t count = 0;
            for(uint32_t x = 0; x < nx[t]; x++) {
                const auto it = where.find(kmerIds[begin0[t] + x]);
                if(it == where.end()) continue;
                count += it->second.size();
            }
            // (a second pass places them: scattered over the room by a stride coprime to the count, so that the list is in no order)
            const uint64_t stored = std::min<uint64_t>(count, capacity);
            uint64_t stride = 7919; while(stored && std::__gcd(stride, stored) != 1) ++stride;
            uint64_t at = 0;
            for(uint32_t x = 0; x < nx[t] && stored; x++) {
                const auto it = where.find(kmerIds[begin0[t] + x]);
                if(it == where.end()) continue;
                for(const uint32_t y : it->second) { if(at < stored) hostHits[hitBase[t] + (at * stride) % stored] = (x << 16) | y; ++at; }
            }
            meta[t] = uint32_t(std::min<uint64_t>(count, 0x7fffffffu)) | ((t & 1) ? 0x80000000u : 0u);
        }
        b.hits.reserve(hostHits.size(), stream); b.hitBase.reserve(hitBase.size(), stream); b.hitMeta.reserve(taskCount, stream);
        HIP_CHECK(hipMemcpyAsync(b.hits.data(), hostHits.data(), hostHits.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(b.hitBase.data(), hitBase.data(), hitBase.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(b.hitMeta.data(), meta.data(), taskCount * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        uint32_t longest = 0;
        for(uint64_t t = 0; t < taskCount; t++) longest = std::max(longest, std::max(nx[t], ny[t]));
        sparseInput = SparseInput{b.hits.data(), b.hitBase.data(), b.hitMeta.data(), longest};
    }
    ev.create();
    try { (void)runDpTasks(ctx, ws, b, narrowCount, opt, &ev, &stats, nullptr, &wideTasks, &pairs, sparse ? &sparseInput : nullptr); } catch(...) { ev.destroy(); throw; }
    std::vector<DpResult> results(taskCount);
    HIP_CHECK(hipMemcpyAsync(results.data(), b.results.data(), taskCount * sizeof(DpResult), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    ev.destroy();
    // HIP-event time of the forward launch of every band class and of the traceback (for kernel A/B runs).
    for(const KernelTimers::Entry& e : ctx.timers.table()) {
        for(int c = 0; c < DP_CLASSES; c++) if(e.name == DP_FORWARD_NAMES[c]) { if(seconds) seconds[c] = e.seconds; if(cells) cells[c] = stats.cells[c]; }
        if(e.name.rfind("dpTraceback", 0) == 0 && seconds) seconds[DP_CLASSES] += e.seconds;
    }
    std::vector<uint32_t> entryOf(taskCount);
    for(uint32_t k = 0; k < taskCount; k++) entryOf[position[k]] = k;
    uint64_t used = 0;
    for(uint64_t t = 0; t < taskCount; t++) {
        const DpResult& r = results[entryOf[t]];
        if(ordinals && used + r.markerCount > capacity) throw std::runtime_error("banded_dp_many: output capacity too small.");
        if(ordinals && r.markerCount) HIP_CHECK(hipMemcpy(ordinals + 2 * used, b.ordScratch.data() + 2 * r.ordBegin, 8ULL * r.markerCount, hipMemcpyDeviceToHost));
        counts[t] = r.markerCount;
        scores[t] = r.score;
        used += r.markerCount;
    }
}

void bandedDpUnit(const uint32_t* k0, uint32_t nx, const uint32_t* k1, uint32_t ny, int32_t bandMin, int32_t bandMax,
    uint32_t* ordinals, uint64_t capacity, uint64_t* count, int32_t* score)
{
    int device = 0;
    HIP_CHECK(hipGetDevice(&device));
    Context ctx(device);
    if(bandMin > bandMax || bandMax - bandMin + 1 > 1024) throw std::runtime_error("banded_dp: band width must be in [1, 1024].");
    if(bandMin > int32_t(nx) || bandMax < -int32_t(ny) || nx == 0 || ny == 0) {
        *count = 0; *score = int32_t(0x80000000);
        return;
    }
    std::vector<uint64_t> toc = {0, nx, uint64_t(nx) + ny};
    std::vector<uint32_t> all(k0, k0 + nx);
    all.insert(all.end(), k1, k1 + ny);
    // A context with one "read" whose two strands are the two sequences.
    ctx.setMarkers(1, toc.data(), nullptr, all.data(), nullptr);
    hipStream_t stream = ctx.stream;
    BatchScratch b;
    PairDesc pd; pd.begin0 = 0; pd.begin1 = nx; pd.nx = nx; pd.ny = ny;
    DpTask task; task.pair = 0; task.bandMin = bandMin; task.bandMax = bandMax; task.label = 0;
    b.pairs.reserve(1, stream); b.tasks.reserve(1, stream); b.layoutZeroed(1, stream);
    HIP_CHECK(hipMemcpyAsync(b.pairs.data(), &pd, sizeof(pd), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(b.tasks.data(), &task, sizeof(task), hipMemcpyHostToDevice, stream));

    DeviceOptions opt;
    std::memset(&opt, 0, sizeof(opt));
    opt.deltaX = 200; opt.deltaY = 10; opt.maxSkip = opt.maxDrift = opt.maxTrim = ~0ULL; opt.maxBand = 1024;
    const WorkStream ws{ctx.stream, &ctx.sortWs, nullptr};
    (void)runDpTasks(ctx, ws, b, 1, opt, nullptr, nullptr);
    DpResult r;
    HIP_CHECK(hipMemcpyAsync(&r, b.results.data(), sizeof(r), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if(r.markerCount > capacity) throw std::runtime_error("banded_dp: output capacity too small.");
    if(r.markerCount) {
        HIP_CHECK(hipMemcpy(ordinals, b.ordScratch.data() + 2 * r.ordBegin, 8ULL * r.markerCount, hipMemcpyDeviceToHost));
    }
    *count = r.markerCount;
    *score = r.score;
}

}  // namespace shasta_mi355x
