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
#include "context.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <set>
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
    uint32_t pad;
};

struct DeviceOptions {
    uint32_t deltaX, deltaY;
    uint64_t minEntryCountPerCell, maxDistanceFromBoundary, minAlignedMarkerCount;
    double minAlignedFraction;
    uint64_t maxSkip, maxDrift, maxTrim, maxBand;
    uint32_t suppressContainments;
};

// ---------------------------------------------------------------------------
// K8/K9: cells.
// ---------------------------------------------------------------------------
constexpr int CELLS_THREADS = 256;
constexpr int MATCH_CHUNK = 2048;          // markers of read 1 hashed per round
constexpr int MATCH_SLOTS = 4096;
constexpr int CELL_SLOTS = 2048;
constexpr int MAX_CELLS = 1024;
constexpr uint32_t EMPTY32 = 0xffffffffu;
constexpr uint64_t EMPTY64 = ~0ULL;

constexpr uint32_t F_NEAR_LT = 1, F_NEAR_RB = 2, F_FWD = 4, F_BWD = 8;
constexpr uint8_t PAIR_RESOURCE = 1;       // a cell table overflowed: retried with a larger table in HBM
constexpr uint8_t PAIR_TOO_LONG = 2;       // outside the supported geometry (iX/iY >= 2^16 or band > 1024): skipped + reported

__device__ __forceinline__ uint32_t hash32(uint32_t k) { return k * 2654435761u; }

// getxy, src/Align4.cpp:184-191 (int32, C++ truncating division).
__device__ __forceinline__ void getxy(uint32_t X, uint32_t Y, uint32_t nx, int32_t& x, int32_t& y)
{
    const int32_t Xs = int32_t(X), Ys = int32_t(Y);
    x = (Xs - Ys + int32_t(nx) - 1) / 2;
    y = (Xs + Ys - int32_t(nx) + 1) / 2;
}

// Reads of block-shared mutable state.  SMALL: LDS.  BIG: the tables live in HBM scratch and
// are updated with atomics (L2); plain loads could hit a stale line of this CU's L1, so
// they bypass it (agent-scope relaxed load = global_load sc1).
template<bool BIG> __device__ __forceinline__ uint32_t ld(const uint32_t* p)
{
    if(BIG) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// One workgroup per candidate.  SMALL keeps the cell table (CELL_SLOTS) and the kept-cell
// list (MAX_CELLS) in LDS; BIG uses a per-candidate region of HBM scratch of 2^slotsLog2
// table slots (layout: keys[S] vals[S] cKey[S/2] cFlags[S/2] cLabel[S/2] cYMin[S/2] cYMax[S/2]).
template<bool BIG>
__global__ void __launch_bounds__(CELLS_THREADS)
align4CellsKernel(
    const uint32_t* __restrict__ kmerIds, const PairDesc* __restrict__ pairs,
    const uint32_t* __restrict__ pairList, uint32_t listCount,
    DeviceOptions opt, DpTask* __restrict__ tasks, uint32_t* __restrict__ taskCount, uint32_t taskCapacity,
    uint8_t* __restrict__ pairFlags,
    uint32_t* __restrict__ bigScratch, const uint64_t* __restrict__ bigOffsets, const uint8_t* __restrict__ bigSlotsLog2)
{
    __shared__ uint64_t matchTab[MATCH_SLOTS];
    __shared__ uint32_t sCellKeys[BIG ? 1 : CELL_SLOTS];
    __shared__ uint32_t sCellVals[BIG ? 1 : CELL_SLOTS];
    __shared__ uint32_t sKey[BIG ? 1 : MAX_CELLS];
    __shared__ uint32_t sFlags[BIG ? 1 : MAX_CELLS];
    __shared__ uint32_t sLabel[BIG ? 1 : MAX_CELLS];
    __shared__ uint32_t sYMin[BIG ? 1 : MAX_CELLS];
    __shared__ uint32_t sYMax[BIG ? 1 : MAX_CELLS];
    __shared__ uint32_t sCells, sOverflow, sChanged;

    if(blockIdx.x >= listCount) return;
    const uint32_t pair = pairList[blockIdx.x];
    const int tid = int(threadIdx.x);
    const PairDesc pd = pairs[pair];
    const uint32_t nx = pd.nx, ny = pd.ny;
    const uint32_t* __restrict__ p0 = kmerIds + pd.begin0;
    const uint32_t* __restrict__ p1 = kmerIds + pd.begin1;

    uint32_t *cellKeys, *cellVals, *cKey, *cFlags, *cLabel, *cYMin, *cYMax;
    int slotsLog2;
    if(BIG) {
        slotsLog2 = int(bigSlotsLog2[blockIdx.x]);
        const uint64_t S = 1ULL << slotsLog2;
        uint32_t* base = bigScratch + bigOffsets[blockIdx.x];
        cellKeys = base; cellVals = base + S; cKey = base + 2 * S; cFlags = cKey + S / 2;
        cLabel = cFlags + S / 2; cYMin = cLabel + S / 2; cYMax = cYMin + S / 2;
    } else {
        slotsLog2 = 11;
        cellKeys = sCellKeys; cellVals = sCellVals; cKey = sKey; cFlags = sFlags; cLabel = sLabel; cYMin = sYMin; cYMax = sYMax;
    }
    const uint32_t slots = 1u << slotsLog2;
    const uint32_t maxCells = BIG ? slots / 2 : uint32_t(MAX_CELLS);
    const int hashShift = 32 - slotsLog2;

    for(uint32_t k = tid; k < slots; k += CELLS_THREADS) { cellKeys[k] = EMPTY32; cellVals[k] = 0; }
    if(tid == 0) { sCells = 0; sOverflow = 0; sChanged = 0; }

    // --- alignment matrix entries -> per-cell counts (createAlignmentMatrix + createCells) ---
    for(uint32_t chunk = 0; chunk < ny; chunk += MATCH_CHUNK) {
        __syncthreads();
        for(int k = tid; k < MATCH_SLOTS; k += CELLS_THREADS) matchTab[k] = EMPTY64;
        __syncthreads();
        const uint32_t chunkEnd = min(ny, chunk + uint32_t(MATCH_CHUNK));
        for(uint32_t y = chunk + tid; y < chunkEnd; y += CELLS_THREADS) {
            const uint32_t k = p1[y];
            const unsigned long long entry = (uint64_t(k) << 32) | y;
            uint32_t slot = hash32(k) >> (32 - 12);
            for(;;) {
                const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&matchTab[slot]), EMPTY64, entry);
                if(old == EMPTY64) break;
                slot = (slot + 1) & (MATCH_SLOTS - 1);
            }
        }
        __syncthreads();
        for(uint32_t x = tid; x < nx; x += CELLS_THREADS) {
            const uint32_t k = p0[x];
            uint32_t slot = hash32(k) >> (32 - 12);
            for(;;) {
                const uint64_t e = matchTab[slot];
                if(e == EMPTY64) break;
                if(uint32_t(e >> 32) == k) {
                    const uint32_t y = uint32_t(e);
                    const uint32_t X = x + y, Y = nx + y - x - 1;                  // getXY, :171-177
                    const uint32_t iX = X / opt.deltaX, iY = Y / opt.deltaY;
                    if(iX >= 65536u || iY >= 65535u) { sOverflow = 2; }
                    else {
                        const uint32_t key = (iY << 16) | iX;
                        uint32_t cs = hash32(key) >> hashShift;
                        uint32_t probe = 0;
                        for(; probe < slots; probe++) {
                            const uint32_t old = atomicCAS(&cellKeys[cs], EMPTY32, key);
                            if(old == EMPTY32 || old == key) { atomicAdd(&cellVals[cs], 1u); break; }
                            cs = (cs + 1) & (slots - 1);
                        }
                        if(probe == slots) sOverflow = 1;
                    }
                }
                slot = (slot + 1) & (MATCH_SLOTS - 1);
            }
        }
    }
    __syncthreads();

    // Keep cells with enough entries (:417) and give them compact indices.
    for(uint32_t k = tid; k < slots; k += CELLS_THREADS) {
        const uint32_t key = ld<BIG>(&cellKeys[k]);
        if(key == EMPTY32) continue;
        if(uint64_t(ld<BIG>(&cellVals[k])) >= opt.minEntryCountPerCell) {
            const uint32_t idx = atomicAdd(&sCells, 1u);
            if(idx < maxCells) { cKey[idx] = key; cellVals[k] = idx; }
            else { sOverflow = 1; cellVals[k] = EMPTY32; }
        } else {
            cellVals[k] = EMPTY32;
        }
    }
    __syncthreads();
    if(sOverflow) { if(tid == 0) pairFlags[pair] = (sOverflow == 2) ? PAIR_TOO_LONG : PAIR_RESOURCE; return; }
    const int n = int(sCells);
    if(n == 0) return;

    auto find = [&](int32_t iX, int32_t iY) -> int {
        if(iX < 0 || iY < 0 || iX >= 65536 || iY >= 65535) return -1;
        const uint32_t key = (uint32_t(iY) << 16) | uint32_t(iX);
        uint32_t cs = hash32(key) >> hashShift;
        for(uint32_t probe = 0; probe < slots; probe++) {
            const uint32_t k = ld<BIG>(&cellKeys[cs]);
            if(k == EMPTY32) return -1;
            if(k == key) return int(ld<BIG>(&cellVals[cs]));          // EMPTY32 (-1) for dropped cells
            cs = (cs + 1) & (slots - 1);
        }
        return -1;
    };

    // Boundary flags (:424-429 with the corner rules of :530-626).
    for(int c = tid; c < n; c += CELLS_THREADS) {
        const uint32_t key = ld<BIG>(&cKey[c]);
        const uint32_t iX = key & 0xffffu, iY = key >> 16;
        int32_t x, y;
        getxy(iX * opt.deltaX, (iY + 1) * opt.deltaY, nx, x, y);
        const uint32_t left = x < 0 ? 0u : uint32_t(x);
        getxy((iX + 1) * opt.deltaX, iY * opt.deltaY, nx, x, y);
        const uint32_t right = (x >= int32_t(nx) - 1) ? 0u : uint32_t(nx - 1 - uint32_t(x));
        getxy(iX * opt.deltaX, iY * opt.deltaY, nx, x, y);
        const uint32_t top = y < 0 ? 0u : uint32_t(y);
        getxy((iX + 1) * opt.deltaX, (iY + 1) * opt.deltaY, nx, x, y);
        const uint32_t bottom = (y >= int32_t(ny) - 1) ? 0u : uint32_t(ny - 1 - uint32_t(y));
        uint32_t f = 0;
        if(uint64_t(left) < opt.maxDistanceFromBoundary || uint64_t(top) < opt.maxDistanceFromBoundary) f |= F_NEAR_LT | F_FWD;
        if(uint64_t(right) < opt.maxDistanceFromBoundary || uint64_t(bottom) < opt.maxDistanceFromBoundary) f |= F_NEAR_RB;
        cFlags[c] = f;
        cYMin[c] = EMPTY32; cYMax[c] = 0;
    }

    // forwardSearch (:682-729): a cell is forward accessible if a forward accessible cell
    // lies at (iX-1 or iX, iY-1..iY+1).  Label propagation to the fixed point.
    for(;;) {
        __syncthreads();
        if(tid == 0) sChanged = 0;
        __syncthreads();
        for(int c = tid; c < n; c += CELLS_THREADS) {
            if(ld<BIG>(&cFlags[c]) & F_FWD) continue;
            const uint32_t key = ld<BIG>(&cKey[c]);
            const int32_t iX = int32_t(key & 0xffffu), iY = int32_t(key >> 16);
            bool reach = false;
            for(int dY = -1; dY <= 1 && !reach; dY++) for(int dX = -1; dX <= 0; dX++) {
                if(dX == 0 && dY == 0) continue;
                const int j = find(iX + dX, iY + dY);
                if(j >= 0 && (ld<BIG>(&cFlags[j]) & F_FWD)) { reach = true; break; }
            }
            if(reach) { atomicOr(&cFlags[c], F_FWD); sChanged = 1; }
        }
        __syncthreads();
        if(!sChanged) break;
    }
    // backwardSearch (:736-787): seeds near right/bottom AND forward accessible; a cell is
    // backward accessible if a backward accessible cell lies at (iX or iX+1, iY-1..iY+1).
    for(int c = tid; c < n; c += CELLS_THREADS) {
        const uint32_t f = ld<BIG>(&cFlags[c]);
        if((f & F_NEAR_RB) && (f & F_FWD)) atomicOr(&cFlags[c], F_BWD);
    }
    for(;;) {
        __syncthreads();
        if(tid == 0) sChanged = 0;
        __syncthreads();
        for(int c = tid; c < n; c += CELLS_THREADS) {
            if(ld<BIG>(&cFlags[c]) & F_BWD) continue;
            const uint32_t key = ld<BIG>(&cKey[c]);
            const int32_t iX = int32_t(key & 0xffffu), iY = int32_t(key >> 16);
            bool reach = false;
            for(int dY = -1; dY <= 1 && !reach; dY++) for(int dX = 0; dX <= 1; dX++) {
                if(dX == 0 && dY == 0) continue;
                const int j = find(iX + dX, iY + dY);
                if(j >= 0 && (ld<BIG>(&cFlags[j]) & F_BWD)) { reach = true; break; }
            }
            if(reach) { atomicOr(&cFlags[c], F_BWD); sChanged = 1; }
        }
        __syncthreads();
        if(!sChanged) break;
    }

    // Connected components of active cells, 8-neighbourhood (:792-868): min-label propagation.
    for(int c = tid; c < n; c += CELLS_THREADS) {
        const uint32_t f = ld<BIG>(&cFlags[c]);
        cLabel[c] = ((f & F_FWD) && (f & F_BWD)) ? ld<BIG>(&cKey[c]) : EMPTY32;
    }
    for(;;) {
        __syncthreads();
        if(tid == 0) sChanged = 0;
        __syncthreads();
        for(int c = tid; c < n; c += CELLS_THREADS) {
            const uint32_t mine = ld<BIG>(&cLabel[c]);
            if(mine == EMPTY32) continue;
            const uint32_t key = ld<BIG>(&cKey[c]);
            const int32_t iX = int32_t(key & 0xffffu), iY = int32_t(key >> 16);
            uint32_t best = mine;
            for(int dY = -1; dY <= 1; dY++) for(int dX = -1; dX <= 1; dX++) {
                if(dX == 0 && dY == 0) continue;
                const int j = find(iX + dX, iY + dY);
                if(j >= 0) best = min(best, ld<BIG>(&cLabel[j]));
            }
            if(best < mine) { atomicMin(&cLabel[c], best); sChanged = 1; }
        }
        __syncthreads();
        if(!sChanged) break;
    }
    // iY range of each component, stored at its root cell (the cell whose key is the label).
    for(int c = tid; c < n; c += CELLS_THREADS) {
        const uint32_t label = ld<BIG>(&cLabel[c]);
        if(label == EMPTY32) continue;
        const int r = find(int32_t(label & 0xffffu), int32_t(label >> 16));
        const uint32_t iY = ld<BIG>(&cKey[c]) >> 16;
        atomicMin(&cYMin[r], iY);
        atomicMax(&cYMax[r], iY);
    }
    __syncthreads();
    // One banded alignment per component (:890-934).
    for(int c = tid; c < n; c += CELLS_THREADS) {
        const uint32_t key = ld<BIG>(&cKey[c]);
        if(ld<BIG>(&cLabel[c]) != key) continue;
        const uint32_t YMin = ld<BIG>(&cYMin[c]) * opt.deltaY;
        const uint32_t YMax = (ld<BIG>(&cYMax[c]) + 1) * opt.deltaY - 1;
        const int32_t bandMin = int32_t(nx) - 1 - int32_t(YMax);
        const int32_t bandMax = int32_t(nx) - 1 - int32_t(YMin);
        const int32_t bandWidth = bandMax - bandMin + 1;
        if(int64_t(bandWidth) > int64_t(opt.maxBand)) continue;             // :929
        if(bandWidth > 1024) { pairFlags[pair] = PAIR_TOO_LONG; continue; }
        const uint32_t t = atomicAdd(taskCount, 1u);
        if(t < taskCapacity) { DpTask task; task.pair = pair; task.bandMin = bandMin; task.bandMax = bandMax; task.label = key; tasks[t] = task; }
    }
}

// ---------------------------------------------------------------------------
// K8/K9, fast path.  A CHUNK is a set of candidates that share one oriented read: read 0
// (candidates arrive sorted by readId0, src/LowHash0.cpp:204-214, and read 0 is always on
// strand 0, src/AssemblerAlign.cpp:382) or, when read 1 is the shorter one, read 1 ("swapped",
// gathered by the host).  One workgroup per chunk; its waves share the table of that read and
// then work on different candidates of the chunk without ever synchronising again.
//   build   read 0's kmer ids are copied to LDS and indexed by a two-choice bucketised LDS hash
//           table (buckets of four 16-bit slots = one ds_read_b64; slot = hash tag | ordinal);
//   probe   a candidate's other read is streamed through the table, four markers per lane per
//           round: both buckets are read, the eight slots are tag-matched with SWAR compares,
//           the kmer ids are compared in LDS: fixed trip count, no probe chains;
//   count   (x,y) -> cell by magic-number division (getXY + createCells,
//           src/Align4.cpp:171-177,380-436), one LDS atomic per hit on a packed cell word (folding
//           equal neighbours first costs more instructions than the atomics it saves); the
//           increment that reaches minEntryCountPerCell appends the cell to the kept list (:417);
//   graph   the kept cells (Q per lane) live in registers; their forward/backward adjacency is
//           a bit mask per cell, so forwardSearch / backwardSearch (:682-788) and the connected
//           components (:792-868) are iterated ballots with no memory traffic;
//   tasks   one DP task per component (:890-934), staged in LDS, appended with one global atomic.
// Candidates that overflow a table or the kept list are flagged PAIR_RESOURCE and retried in a
// larger class, finally by align4CellsKernel<true>.
// Dynamic LDS (32-bit words): aKmers[NA] | aSlots[NA] (2 NA 16-bit slots) | per wave:
//   cells[SC] (iY | iX | count packed) | kept[64 Q] | scratch[8] | stage[4 CELLS_STAGE].
// ---------------------------------------------------------------------------
// firstMember indexes the member list (candidate indices of the batch).
struct CellsChunk { uint32_t firstMember; uint16_t count, swapped; uint32_t naLog2, scLog2; };

#ifdef SHASTA_PROFILE_PHASES
__device__ unsigned long long g_phaseCycles[16];
#define PHASE_MARK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
    if((threadIdx.x & 63) == 0) atomicAdd(&g_phaseCycles[k], now_ - phaseT_); phaseT_ = now_; } while(0)
#define PHASE_BEGIN() unsigned long long phaseT_ = __builtin_readcyclecounter()
#else
#define PHASE_MARK(k) do {} while(0)
#define PHASE_BEGIN() do {} while(0)
#endif

// floor(v / d) = umulhi(v, magic) with magic = floor(2^32 / d) + 1, exact whenever v * d < 2^32
// (the host only sends a candidate to this kernel if (nx + ny) * max(deltaX, deltaY) < 2^32).
__device__ __forceinline__ uint32_t divMagic(uint32_t v, uint32_t magic) { return __umulhi(v, magic); }

// LDS traffic of ONE wave is ordered by the hardware; this only stops the compiler from moving
// LDS accesses across it and drains the counters.  Waves of a chunk never wait for each other
// after the build, so no s_barrier may appear in the per-candidate code.
__device__ __forceinline__ void waveLdsSync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t hash32b(uint32_t k) { return k * 0x85ebca6bu; }

// Bits 15 and 31 of the result flag the 16-bit halves of v that are zero (exact, no carries).
__device__ __forceinline__ uint32_t zeroHalves(uint32_t v)
{
    return ~(((v & 0x7fff7fffu) + 0x7fff7fffu) | v | 0x7fff7fffu);
}

constexpr int CELLS_UNROLL = 4;           // markers per lane per round
constexpr int CELLS_IX_BITS = 10, CELLS_IY_BITS = 12, CELLS_COUNT_BITS = 10;   // packed LDS cell word
constexpr int CELLS_STAGE = 8;            // DP tasks staged per wave before one global append
__host__ __device__ inline size_t cellsWaveLdsWords(int scLog2, int Q)
{
    return (size_t(1) << scLog2) + 64 * size_t(Q) + 8 + 4 * CELLS_STAGE;
}
__host__ __device__ inline size_t cellsChunkLdsWords(int naLog2, int scLog2, int Q, int waves)
{
    return 2 * (size_t(1) << naLog2) + size_t(waves) * cellsWaveLdsWords(scLog2, Q);
}

template<int Q>
__global__ void __launch_bounds__(384)
align4CellsChunkKernel(
    const uint32_t* __restrict__ kmerIds, const PairDesc* __restrict__ pairs,
    const CellsChunk* __restrict__ chunks, uint32_t chunkCount, const uint32_t* __restrict__ members,
    DeviceOptions opt, uint32_t magicX, uint32_t magicY,
    DpTask* __restrict__ tasks, uint32_t* __restrict__ taskCount, uint32_t taskCapacity,
    uint8_t* __restrict__ pairFlags)
{
    extern __shared__ uint32_t ldsWords[];
    // Markers whose two buckets were both full when they arrived (the two-choice table runs at two
    // entries per four-slot bucket on average, so a nearly full table overflows now and then).
    constexpr uint32_t STASH = 32;
    __shared__ uint32_t stashCount, stashKmer[STASH], stashOrdinal[STASH];
    constexpr int MAXC = 64 * Q;
    if(blockIdx.x >= chunkCount) return;
    const CellsChunk chunk = chunks[blockIdx.x];
    const int lane = laneId();
    const uint32_t wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint32_t NA = 1u << chunk.naLog2, SC = 1u << chunk.scLog2;
    const int bucketShift = 32 - (int(chunk.naLog2) - 1), scShift = 32 - int(chunk.scLog2);
    const int xBits = int(chunk.naLog2);
    const uint32_t xMask = NA - 1, tagMask = (1u << (16 - xBits)) - 1;
    uint32_t* const aKmers = ldsWords;
    uint32_t* const aSlots = aKmers + NA;
    uint32_t* const cells = aSlots + NA + wave * cellsWaveLdsWords(int(chunk.scLog2), Q);
    uint32_t* const kept = cells + SC;
    uint32_t* const scratch = kept + MAXC;                        // [0] kept count, [1] min, [2] max, [3] staged tasks
    uint32_t* const stage = scratch + 8;
    const uint32_t threshold = uint32_t(opt.minEntryCountPerCell > 1 ? min(opt.minEntryCountPerCell, uint64_t(0xffffffffu)) : 1);
    PHASE_BEGIN();

    // Tag of a kmer id (never all ones, so that an empty slot matches no tag) and its two buckets.
    auto tagOf = [&](uint32_t h) { const uint32_t t = (h >> 4) & tagMask; return t == tagMask ? 0u : t; };

    // --- table of the read shared by every candidate of the chunk: read 0, or (swapped chunk:
    //     candidates gathered by the host because they share a short read 1) read 1 ---
    const PairDesc pdFirst = pairs[members[chunk.firstMember]];
    const bool swapped = chunk.swapped != 0;
    const uint32_t* __restrict__ tabSeq = kmerIds + (swapped ? pdFirst.begin1 : pdFirst.begin0);
    const uint32_t tabCount = swapped ? pdFirst.ny : pdFirst.nx;  // < NA (host)
    for(uint32_t k = threadIdx.x; k < NA; k += blockDim.x) aSlots[k] = 0xffffffffu;
    if(threadIdx.x == 0) stashCount = 0;
    if(lane == 0) scratch[3] = 0;
    __syncthreads();
    for(uint32_t t = threadIdx.x; t < tabCount; t += blockDim.x) {
        const uint32_t km = tabSeq[t];
        aKmers[t] = km;
        const uint32_t h = hash32(km);
        const uint32_t b1 = h >> bucketShift, b2 = hash32b(km) >> bucketShift;
        const uint32_t entry = (tagOf(h) << xBits) | t;
        for(;;) {
            // Free slots of the two candidate buckets; take the emptier bucket (ties: the first).
            const uint32_t u0 = aSlots[2 * b1], u1 = aSlots[2 * b1 + 1], v0 = aSlots[2 * b2], v1 = aSlots[2 * b2 + 1];
            const uint32_t fu0 = zeroHalves(~u0), fu1 = zeroHalves(~u1), fv0 = zeroHalves(~v0), fv1 = zeroHalves(~v1);
            const int freeU = __popc(fu0) + __popc(fu1), freeV = __popc(fv0) + __popc(fv1);
            if(freeU == 0 && freeV == 0) {
                const uint32_t k = atomicAdd(&stashCount, 1u);
                if(k < STASH) { stashKmer[k] = km; stashOrdinal[k] = t; }
                break;
            }
            const bool useV = freeV > freeU;
            const uint32_t f0 = useV ? fv0 : fu0, f1 = useV ? fv1 : fu1;
            const uint32_t w0 = useV ? v0 : u0, w1 = useV ? v1 : u1;
            const uint32_t base = 2 * (useV ? b2 : b1);
            const bool second = f0 == 0;
            const uint32_t f = second ? f1 : f0, old = second ? w1 : w0;
            const int shift = (f & 0x8000u) ? 0 : 16;
            const uint32_t updated = (old & ~(0xffffu << shift)) | (entry << shift);
            if(atomicCAS(&aSlots[base + (second ? 1 : 0)], old, updated) == old) break;
        }
    }
    __syncthreads();
    PHASE_MARK(0);
    const uint32_t stashed = stashCount;
    if(stashed > STASH) {
        // Too many markers of the tabled read share their buckets (a tandem repeat): the whole
        // chunk goes to the next class.
        for(uint32_t c = threadIdx.x; c < chunk.count; c += blockDim.x) pairFlags[members[chunk.firstMember + c]] = uint8_t(PAIR_RESOURCE | 0x80);
        return;
    }

    for(uint32_t c = wave; c < chunk.count; c += waves) {
        const uint32_t pair = members[chunk.firstMember + c];
        const PairDesc pd = pairs[pair];
        const uint32_t nx = pd.nx, ny = pd.ny;
        const uint32_t* __restrict__ stream = kmerIds + (swapped ? pd.begin0 : pd.begin1);
        const uint32_t streamCount = swapped ? nx : ny;
        int overflow = 0, reason = 0;

        for(uint32_t k = lane; k < SC; k += WAVE) cells[k] = EMPTY32;
        if(lane == 0) scratch[0] = 0;
        waveLdsSync();
        PHASE_MARK(1);

        // --- alignment matrix entries -> per-cell counts (createAlignmentMatrix + createCells) ---
        // Counts the hits of one round: hit[u] with table ordinal ti[u] and stream ordinal t.
        auto countHits = [&](const bool (&hit)[CELLS_UNROLL], const uint32_t (&ti)[CELLS_UNROLL], uint32_t s0) {
            bool pending[CELLS_UNROLL];
            uint32_t key[CELLS_UNROLL], packed[CELLS_UNROLL], cs[CELLS_UNROLL], len[CELLS_UNROLL], probes[CELLS_UNROLL];
#pragma unroll
            for(int u = 0; u < CELLS_UNROLL; u++) {
                const uint32_t t = s0 + u * WAVE + lane;
                const uint32_t x = swapped ? t : ti[u], y = swapped ? ti[u] : t;
                const uint32_t X = x + y, Y = nx + y - x - 1;                  // getXY, :171-177
                const uint32_t iX = divMagic(X, magicX), iY = divMagic(Y, magicY);
                // The LDS cell table packs (iY:12 | iX:10 | count:10) in one word; the host only sends
                // candidates whose cell indices fit (others run in the HBM-scratch kernel).
                const bool h = hit[u];
                key[u] = h ? ((iY << 16) | iX) : EMPTY32;
                len[u] = 1;
                pending[u] = h;
                packed[u] = (iY << CELLS_IX_BITS) | iX;
                cs[u] = hash32(key[u]) >> scShift;
                probes[u] = 0;
            }
            while(__any(pending[0] | pending[1] | pending[2] | pending[3])) {
#pragma unroll
                for(int u = 0; u < CELLS_UNROLL; u++) {
                    if(pending[u]) {
                        const uint32_t cur = *reinterpret_cast<volatile uint32_t*>(&cells[cs[u]]);
                        bool done = false;
                        uint32_t before = 0;
                        if(cur != EMPTY32 && (cur >> CELLS_COUNT_BITS) == packed[u]) {
                            before = atomicAdd(&cells[cs[u]], len[u]) & ((1u << CELLS_COUNT_BITS) - 1);
                            done = true;
                        } else if(cur == EMPTY32) {
                            // Claim the slot; on failure look at the same slot again.
                            done = atomicCAS(&cells[cs[u]], EMPTY32, (packed[u] << CELLS_COUNT_BITS) | len[u]) == EMPTY32;
                        } else {
                            cs[u] = (cs[u] + 1) & (SC - 1);
                            if(++probes[u] == SC) { overflow = max(overflow, 1); reason |= 1; pending[u] = false; }
                        }
                        if(done) {
                            if(before < threshold && before + len[u] >= threshold) {           // :417
                                const uint32_t idx = atomicAdd(&scratch[0], 1u);
                                if(idx < uint32_t(MAXC)) kept[idx] = key[u];
                            }
                            pending[u] = false;
                        }
                    }
                }
            }
        };

        uint32_t kmNext[CELLS_UNROLL];
#pragma unroll
        for(int u = 0; u < CELLS_UNROLL; u++) { const uint32_t t = u * WAVE + lane; kmNext[u] = t < streamCount ? stream[t] : 0u; }
        for(uint32_t s0 = 0; s0 < streamCount; s0 += CELLS_UNROLL * WAVE) {
            uint32_t km[CELLS_UNROLL], w[CELLS_UNROLL][4], m[CELLS_UNROLL][4], ti[CELLS_UNROLL], ka[CELLS_UNROLL];
            bool valid[CELLS_UNROLL], hit[CELLS_UNROLL];
#pragma unroll
            for(int u = 0; u < CELLS_UNROLL; u++) {
                km[u] = kmNext[u];
                const uint32_t tn = s0 + (CELLS_UNROLL + u) * WAVE + lane;
                kmNext[u] = tn < streamCount ? stream[tn] : 0u;                  // prefetch the next round
                valid[u] = s0 + u * WAVE + lane < streamCount;
            }
#pragma unroll
            for(int u = 0; u < CELLS_UNROLL; u++) {
                const uint32_t h = hash32(km[u]);
                const uint32_t b1 = h >> bucketShift, b2 = hash32b(km[u]) >> bucketShift;
                w[u][0] = aSlots[2 * b1]; w[u][1] = aSlots[2 * b1 + 1];
                w[u][2] = aSlots[2 * b2]; w[u][3] = aSlots[2 * b2 + 1];
                const uint32_t pattern = (tagOf(h) << xBits) * 0x00010001u, fieldMask = (tagMask << xBits) * 0x00010001u;
                const bool same = b1 == b2;
#pragma unroll
                for(int i = 0; i < 4; i++) m[u][i] = zeroHalves((w[u][i] ^ pattern) & fieldMask);
                if(same) { m[u][2] = 0; m[u][3] = 0; }
                if(!valid[u]) { m[u][0] = m[u][1] = m[u][2] = m[u][3] = 0; }
            }
            // Resolve the tag matches (usually one per marker) against the kmer ids in LDS.
            for(;;) {
                bool more = false;
#pragma unroll
                for(int u = 0; u < CELLS_UNROLL; u++) {
                    // First tag match of this marker (static register indexing only).
                    const bool s0m = m[u][0] != 0, s1m = !s0m && m[u][1] != 0, s2m = !s0m && !s1m && m[u][2] != 0;
                    const uint32_t mm = s0m ? m[u][0] : (s1m ? m[u][1] : (s2m ? m[u][2] : m[u][3]));
                    const uint32_t ww = s0m ? w[u][0] : (s1m ? w[u][1] : (s2m ? w[u][2] : w[u][3]));
                    const bool cand = mm != 0;
                    const bool low = (mm & 0x8000u) != 0;
                    ti[u] = (low ? ww : (ww >> 16)) & xMask;
                    ka[u] = aKmers[cand ? ti[u] : 0u];
                    const uint32_t cleared = mm & (low ? ~0x8000u : ~0x80000000u);
                    if(s0m) m[u][0] = cleared; else if(s1m) m[u][1] = cleared; else if(s2m) m[u][2] = cleared; else m[u][3] = cleared;
                    hit[u] = cand;
                    more |= (m[u][0] | m[u][1] | m[u][2] | m[u][3]) != 0;
                }
                bool anyHit = false;
#pragma unroll
                for(int u = 0; u < CELLS_UNROLL; u++) { hit[u] = hit[u] && ka[u] == km[u]; anyHit |= hit[u]; }
                if(__any(anyHit)) countHits(hit, ti, s0);
                if(!__any(more)) break;
            }
            // The few markers that did not fit their buckets.
            for(uint32_t k = 0; k < stashed; k++) {
                const uint32_t sk = stashKmer[k], so = stashOrdinal[k];
                bool anyHit = false;
#pragma unroll
                for(int u = 0; u < CELLS_UNROLL; u++) { hit[u] = valid[u] && km[u] == sk; ti[u] = so; anyHit |= hit[u]; }
                if(__any(anyHit)) countHits(hit, ti, s0);
            }
        }
        waveLdsSync();
        PHASE_MARK(2);

        const int n = int(scratch[0]);
        if(n > MAXC) { overflow = max(overflow, 1); reason |= 2; }
        const uint64_t anyHard = __ballot(overflow == 2), anySoft = __ballot(overflow == 1);
        if(anyHard || anySoft) {
            // Bits 4-6 carry the reason (cell table full / kept list full / geometry) for diagnostics.
            const int reasons = (__ballot(reason & 1) ? 1 : 0) | (__ballot(reason & 2) ? 2 : 0) | (__ballot(reason & 4) ? 4 : 0);
            if(lane == 0) pairFlags[pair] = anyHard ? PAIR_TOO_LONG : uint8_t(PAIR_RESOURCE | (reasons << 4));
            continue;
        }
        if(n == 0) continue;
        const int nq = (n + WAVE - 1) / WAVE;

        // --- kept cells in registers: boundary flags (:424-429 with the corner rules of :530-626) ---
        uint32_t key[Q], flags[Q];
#pragma unroll
        for(int q = 0; q < Q; q++) {
            const int cc = lane + q * WAVE;
            key[q] = EMPTY32; flags[q] = 0;
            if(cc >= n) continue;
            key[q] = kept[cc];
            const uint32_t iX = key[q] & 0xffffu, iY = key[q] >> 16;
            int32_t x, y;
            getxy(iX * opt.deltaX, (iY + 1) * opt.deltaY, nx, x, y);
            const uint32_t left = x < 0 ? 0u : uint32_t(x);
            getxy((iX + 1) * opt.deltaX, iY * opt.deltaY, nx, x, y);
            const uint32_t right = (x >= int32_t(nx) - 1) ? 0u : uint32_t(nx - 1 - uint32_t(x));
            getxy(iX * opt.deltaX, iY * opt.deltaY, nx, x, y);
            const uint32_t top = y < 0 ? 0u : uint32_t(y);
            getxy((iX + 1) * opt.deltaX, (iY + 1) * opt.deltaY, nx, x, y);
            const uint32_t bottom = (y >= int32_t(ny) - 1) ? 0u : uint32_t(ny - 1 - uint32_t(y));
            if(uint64_t(left) < opt.maxDistanceFromBoundary || uint64_t(top) < opt.maxDistanceFromBoundary) flags[q] |= F_NEAR_LT;
            if(uint64_t(right) < opt.maxDistanceFromBoundary || uint64_t(bottom) < opt.maxDistanceFromBoundary) flags[q] |= F_NEAR_RB;
        }
        // Adjacency masks.  before[q][r] bit j: cell 64 r + j lies at (iX-1 or iX, iY-1..iY+1) of
        // this lane's cell q (a forward move leads from it to this cell); after: (iX or iX+1, ...).
        uint64_t before[Q][Q], after[Q][Q];
#pragma unroll
        for(int q = 0; q < Q; q++)
#pragma unroll
            for(int r = 0; r < Q; r++) { before[q][r] = 0; after[q][r] = 0; }
#pragma unroll
        for(int r = 0; r < Q; r++) {
            if(r >= nq) break;
            const int jEnd = min(WAVE, n - r * WAVE);
            for(int j = 0; j < jEnd; j++) {
                const uint32_t other = __builtin_amdgcn_readlane(key[r], j);
                const int32_t oX = int32_t(other & 0xffffu), oY = int32_t(other >> 16);
                const uint64_t bit = 1ULL << j;
#pragma unroll
                for(int q = 0; q < Q; q++) {
                    if(q >= nq) break;
                    const int32_t dX = oX - int32_t(key[q] & 0xffffu), dY = oY - int32_t(key[q] >> 16);
                    const bool near = key[q] != EMPTY32 && dY >= -1 && dY <= 1 && other != key[q];
                    if(near && (dX == -1 || dX == 0)) before[q][r] |= bit;
                    if(near && (dX == 0 || dX == 1)) after[q][r] |= bit;
                }
            }
        }
        PHASE_MARK(3);

        // forwardSearch (:682-729): seeds near left/top; closure under forward moves.
        uint64_t fwd[Q], bwd[Q];
#pragma unroll
        for(int q = 0; q < Q; q++) fwd[q] = __ballot((flags[q] & F_NEAR_LT) != 0);
        for(;;) {
            bool changed = false;
#pragma unroll
            for(int q = 0; q < Q; q++) {
                if(q >= nq) break;
                uint64_t reach = 0;
#pragma unroll
                for(int r = 0; r < Q; r++) reach |= before[q][r] & fwd[r];
                const uint64_t grown = fwd[q] | __ballot(reach != 0);
                changed |= grown != fwd[q];
                fwd[q] = grown;
            }
            if(!changed) break;
        }
        // backwardSearch (:736-787): seeds near right/bottom AND forward accessible.
#pragma unroll
        for(int q = 0; q < Q; q++) bwd[q] = __ballot((flags[q] & F_NEAR_RB) != 0) & fwd[q];
        for(;;) {
            bool changed = false;
#pragma unroll
            for(int q = 0; q < Q; q++) {
                if(q >= nq) break;
                uint64_t reach = 0;
#pragma unroll
                for(int r = 0; r < Q; r++) reach |= after[q][r] & bwd[r];
                const uint64_t grown = bwd[q] | __ballot(reach != 0);
                changed |= grown != bwd[q];
                bwd[q] = grown;
            }
            if(!changed) break;
        }
        PHASE_MARK(4);
        // Connected components of the active cells, 8-neighbourhood (:792-868), one at a time,
        // seeded at the remaining active cell with the smallest key; one banded alignment per
        // component (:890-934).
        uint64_t remaining[Q];
        bool anyRemaining = false;
#pragma unroll
        for(int q = 0; q < Q; q++) { remaining[q] = fwd[q] & bwd[q]; anyRemaining |= remaining[q] != 0; }
        while(anyRemaining) {
            uint32_t myMin = EMPTY32;
#pragma unroll
            for(int q = 0; q < Q; q++) if((remaining[q] >> lane) & 1ULL) myMin = min(myMin, key[q]);
            if(lane == 0) { scratch[1] = EMPTY32; }
            waveLdsSync();
            if(myMin != EMPTY32) atomicMin(&scratch[1], myMin);
            waveLdsSync();
            const uint32_t seedKey = scratch[1];
            uint64_t comp[Q];
#pragma unroll
            for(int q = 0; q < Q; q++) comp[q] = __ballot(key[q] == seedKey);
            for(;;) {
                bool changed = false;
#pragma unroll
                for(int q = 0; q < Q; q++) {
                    if(q >= nq) break;
                    uint64_t reach = 0;
#pragma unroll
                    for(int r = 0; r < Q; r++) reach |= (before[q][r] | after[q][r]) & comp[r];
                    const uint64_t grown = comp[q] | (__ballot(reach != 0) & remaining[q]);
                    changed |= grown != comp[q];
                    comp[q] = grown;
                }
                if(!changed) break;
            }
            // iY range of the component.
            if(lane == 0) { scratch[1] = EMPTY32; scratch[2] = 0; }
            waveLdsSync();
#pragma unroll
            for(int q = 0; q < Q; q++) {
                if((comp[q] >> lane) & 1ULL) { atomicMin(&scratch[1], key[q] >> 16); atomicMax(&scratch[2], key[q] >> 16); }
            }
            waveLdsSync();
            const uint32_t YMin = scratch[1] * opt.deltaY;
            const uint32_t YMax = (scratch[2] + 1) * opt.deltaY - 1;
            const int32_t bandMin = int32_t(nx) - 1 - int32_t(YMax);
            const int32_t bandMax = int32_t(nx) - 1 - int32_t(YMin);
            const int32_t bandWidth = bandMax - bandMin + 1;
            if(int64_t(bandWidth) <= int64_t(opt.maxBand)) {                      // :929
                if(bandWidth > 1024) { if(lane == 0) pairFlags[pair] = PAIR_TOO_LONG; }
                else {
                    uint32_t staged = scratch[3];
                    if(staged == CELLS_STAGE) {
                        // Staging area full: append it to the task list.
                        uint32_t base = 0;
                        if(lane == 0) base = atomicAdd(taskCount, staged);
                        base = __builtin_amdgcn_readfirstlane(base);
                        for(uint32_t k = lane; k < 4 * staged; k += WAVE) {
                            const uint32_t t = base + k / 4;
                            if(t < taskCapacity) reinterpret_cast<uint32_t*>(tasks)[4ULL * base + k] = stage[k];
                        }
                        waveLdsSync();
                        staged = 0;
                    }
                    if(lane == 0) {
                        stage[4 * staged] = pair; stage[4 * staged + 1] = uint32_t(bandMin);
                        stage[4 * staged + 2] = uint32_t(bandMax); stage[4 * staged + 3] = seedKey;
                        scratch[3] = staged + 1;
                    }
                    waveLdsSync();
                }
            }
            anyRemaining = false;
#pragma unroll
            for(int q = 0; q < Q; q++) { remaining[q] &= ~comp[q]; anyRemaining |= remaining[q] != 0; }
        }
        PHASE_MARK(5);
    }
    // Append this wave's staged tasks.
    waveLdsSync();
    const uint32_t staged = scratch[3];
    if(staged) {
        uint32_t base = 0;
        if(lane == 0) base = atomicAdd(taskCount, staged);
        base = __builtin_amdgcn_readfirstlane(base);
        for(uint32_t k = lane; k < 4 * staged; k += WAVE) {
            const uint32_t t = base + k / 4;
            if(t < taskCapacity) reinterpret_cast<uint32_t*>(tasks)[4ULL * base + k] = stage[k];
        }
    }
    PHASE_MARK(6);
}

// ---------------------------------------------------------------------------
// K10: banded overlap DP (computeBandedAlignment, src/Align4.cpp:993-1088; the SeqAn call it
// wraps is restated -- tie policy as in oracle/banded_dp.hpp).
//
// DP cell (i,j) (i symbols of read 0, j of read 1 consumed) lives on diagonal d = i-j =
// bandMin + b and anti-diagonal s = i+j.  On step s only diagonals with (s+d) even hold a cell;
// its three predecessors are the same diagonal at s-2 (diagonal move), diagonal b-1 at s-1
// (horizontal, from (i-1,j)) and diagonal b+1 at s-1 (vertical, from (i,j-1)).
//
//   forward   bandedDpForwardKernel<G,C>: a task occupies G lanes, each lane owns C adjacent
//             diagonals in registers; narrow bands are packed 64/G tasks to a wavefront (tasks
//             are sorted by class and length first, so bundled tasks have the same trip count).
//             One loop iteration advances two anti-diagonals (all C cells of a lane); the only
//             cross-lane traffic is one shuffle up and one down.  The kmer ids a lane compares
//             slide through register windows fed by one prefetched load per read and iteration.
//             Trace: 2 bits per cell, one __ballot per bit plane, 2C 64-bit words per iteration.
//   end cell  free end gaps: the best border cell = max over the final value of each diagonal,
//             ties to the smallest (i, j) -- a G-lane reduction after the loop.
//   trace     dpTracebackKernel: ONE LANE per task walks its path through the packed trace
//             (64 tasks per wavefront in flight), writes the aligned ordinals and accumulates
//             AlignmentInfo's metrics (src/Alignment.cpp:67-113, :4-31).
// Trace codes: 0 diagonal+equal kmers, 1 diagonal+different, 2 vertical, 3 horizontal.  Tie
// policy: diagonal >= vertical >= horizontal.
// ---------------------------------------------------------------------------
constexpr int DP_CLASSES = 6;
__host__ __device__ inline int dpClassOfWidth(int32_t w) { return w <= 32 ? 0 : (w <= 64 ? 1 : (w <= 128 ? 2 : (w <= 256 ? 3 : (w <= 512 ? 4 : 5)))); }
__host__ __device__ inline int dpLanes(int cls) { return cls == 0 ? 16 : (cls == 1 ? 32 : 64); }          // G
__host__ __device__ inline int dpDiagonals(int cls) { return cls <= 2 ? 2 : (1 << (cls - 1)); }             // C = 2,2,2,4,8,16

struct DpGeometry { int32_t s0; uint32_t iters; int cls; };
__host__ __device__ inline DpGeometry dpGeometry(int32_t bandMin, int32_t bandMax, uint32_t nx, uint32_t ny)
{
    DpGeometry g;
    g.cls = dpClassOfWidth(bandMax - bandMin + 1);
    const int32_t sMin = bandMin > 0 ? bandMin : (bandMax < 0 ? -bandMax : 0);
    g.s0 = sMin - ((sMin + bandMin) & 1);          // (s0 + bandMin) is even
    g.iters = uint32_t((int32_t(nx + ny) - g.s0) / 2 + 1);
    return g;
}

// What the forward kernel leaves for the traceback of a task.
struct DpEnd { uint64_t traceOffset; int32_t bestI, bestJ, score; uint32_t laneBase; };

// Per task: sort key (class, iterations), ordinal capacity, statistics.
__global__ void __launch_bounds__(256)
dpSizeKernel(const DpTask* __restrict__ tasks, const PairDesc* __restrict__ pairs, uint32_t taskCount,
    uint32_t* __restrict__ keys, uint32_t* __restrict__ ids, uint64_t* __restrict__ ordCap,
    uint32_t* __restrict__ classCounts, unsigned long long* __restrict__ sums)   // sums[0] dp cells, sums[1] trace word bound, [2+c] cells of class c, [8+c] bytes of class c
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long cells = 0, words = 0, bytes = 0;
    int cls = -1;
    if(t < taskCount) {
        const DpTask task = tasks[t];
        const PairDesc pd = pairs[task.pair];
        const DpGeometry g = dpGeometry(task.bandMin, task.bandMax, pd.nx, pd.ny);
        cls = g.cls;
        keys[t] = (uint32_t(g.cls) << 24) | min(g.iters, 0xffffffu);
        ids[t] = t;
        ordCap[t] = min(pd.nx, pd.ny);
        cells = (unsigned long long)(pd.nx) * (unsigned long long)(task.bandMax - task.bandMin + 1);
        words = (unsigned long long)(g.iters) * (unsigned long long)(2 * dpDiagonals(g.cls)) + 32;
        bytes = 4ULL * (uint64_t(pd.nx) + pd.ny);
    } else if(t == taskCount) {
        ordCap[t] = 0;
    }
    // Per class: one atomic per wavefront and class present in it (tasks of a wave mostly share a
    // class after the cells kernels); one atomic per task serialises the whole launch on six addresses.
#pragma unroll
    for(int c = 0; c < DP_CLASSES; c++) {
        const uint64_t votes = __ballot(cls == c);
        if(votes == 0) continue;
        unsigned long long classCells = cls == c ? cells : 0, classBytes = cls == c ? bytes : 0;
        for(int d = 32; d >= 1; d >>= 1) { classCells += __shfl_down(classCells, d, WAVE); classBytes += __shfl_down(classBytes, d, WAVE); }
        if(laneId() == 0) {
            atomicAdd(&classCounts[c], uint32_t(__popcll(votes)));
            atomicAdd(&sums[2 + c], classCells);
            atomicAdd(&sums[8 + c], classBytes);
        }
    }
    for(int d = 32; d >= 1; d >>= 1) { cells += __shfl_down(cells, d, WAVE); words += __shfl_down(words, d, WAVE); }
    if(laneId() == 0 && cells) { atomicAdd(&sums[0], cells); atomicAdd(&sums[1], words); }
}

// Trace words of each bundle (64/G consecutive tasks of the sorted list of one class).
struct DpClassLayout { uint32_t taskStart[DP_CLASSES + 1]; uint32_t bundleStart[DP_CLASSES + 1]; };

__global__ void __launch_bounds__(256)
dpBundleKernel(const uint32_t* __restrict__ sortedKeys, DpClassLayout layout, uint64_t* __restrict__ bundleWords)
{
    const uint32_t bundle = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = layout.bundleStart[DP_CLASSES];
    if(bundle > total) return;
    if(bundle == total) { bundleWords[bundle] = 0; return; }
    int cls = 0;
    while(bundle >= layout.bundleStart[cls + 1]) ++cls;
    const uint32_t T = 64u / uint32_t(dpLanes(cls));
    const uint32_t first = layout.taskStart[cls] + (bundle - layout.bundleStart[cls]) * T;
    const uint32_t last = min(first + T, layout.taskStart[cls + 1]) - 1;
    // sorted ascending: the last task has the most iterations.  Rounded to 256 bytes so that the
    // traceback's chunks are whole cache lines.
    bundleWords[bundle] = (uint64_t(sortedKeys[last] & 0xffffffu) * uint64_t(2 * dpDiagonals(cls)) + 31) & ~31ULL;
}

template<int G, int C>
__global__ void __launch_bounds__(256)
bandedDpForwardKernel(
    const uint32_t* __restrict__ kmerIds, const PairDesc* __restrict__ pairs, const DpTask* __restrict__ tasks,
    const uint32_t* __restrict__ sortedIds, uint32_t taskCount,            // this class's segment of the sorted list
    const uint64_t* __restrict__ bundleOffsets, uint32_t bundleCount,      // this class's segment
    uint64_t* __restrict__ trace, DpEnd* __restrict__ ends)
{
    constexpr int T = WAVE / G, HC = C / 2, RW = 2 * C;
    const int lane = laneId();
    const uint32_t bundle = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if(bundle >= bundleCount) return;                     // whole wave leaves; no block barriers below
    const int g = lane / G, l = lane % G;
    const uint32_t pos = bundle * T + uint32_t(g);
    const bool hasTask = pos < taskCount;
    const uint32_t t = sortedIds[hasTask ? pos : bundle * T];
    const DpTask task = tasks[t];
    const PairDesc pd = pairs[task.pair];
    const uint32_t* __restrict__ p0 = kmerIds + pd.begin0;
    const uint32_t* __restrict__ p1 = kmerIds + pd.begin1;
    const int32_t nx = int32_t(pd.nx), ny = int32_t(pd.ny);
    const int32_t bandMin = task.bandMin, width = task.bandMax - task.bandMin + 1;
    const DpGeometry geo = dpGeometry(task.bandMin, task.bandMax, pd.nx, pd.ny);
    uint32_t iters = geo.iters;
#pragma unroll
    for(int d = G; d < WAVE; d <<= 1) iters = max(iters, uint32_t(__shfl_xor(int(iters), d, WAVE)));
    uint64_t* __restrict__ tr = trace + bundleOffsets[bundle];

    // Per diagonal: first and last anti-diagonal that hold a cell of the matrix.
    int32_t lo[C];
    uint32_t span[C];
#pragma unroll
    for(int c = 0; c < C; c++) {
        const int32_t b = l * C + c, d = bandMin + b;
        const int32_t first = d < 0 ? -d : d;
        const int32_t last = min(2 * nx - d, 2 * ny + d);
        const bool exists = hasTask && b < width && d <= nx && d >= -ny && last >= first;
        lo[c] = exists ? first : 0x40000000;
        span[c] = exists ? uint32_t(last - first) : 0u;
    }
    int32_t H[C];
#pragma unroll
    for(int c = 0; c < C; c++) H[c] = NEG_SCORE;

    // Register windows of the kmer ids: aw[k] = A[ib + l HC - 1 + k], bw[h] = B[jb - l HC - 1 - h],
    // ib = (s + bandMin) / 2, jb = ib - bandMin.  Indices are clamped; clamped values belong to
    // cells that are not in the matrix.
    int32_t ib = (geo.s0 + bandMin) / 2;
    auto loadA = [&](int32_t idx) { return p0[min(max(idx, 0), nx - 1)]; };
    auto loadB = [&](int32_t idx) { return p1[min(max(idx, 0), ny - 1)]; };
    uint32_t aw[HC + 1], bw[HC];
#pragma unroll
    for(int k = 0; k <= HC; k++) aw[k] = loadA(ib + l * HC - 1 + k);
#pragma unroll
    for(int h = 0; h < HC; h++) bw[h] = loadB(ib - bandMin - l * HC - 1 - h);
    uint32_t aNext1 = loadA(ib + l * HC + HC), aNext2 = loadA(ib + 1 + l * HC + HC);
    uint32_t bNext1 = loadB(ib - bandMin - l * HC), bNext2 = loadB(ib + 1 - bandMin - l * HC);

    auto cell = [&](int c, int32_t s, uint32_t a, uint32_t bk, int32_t hd, int32_t hv, int32_t hh, uint64_t& loPlane, uint64_t& hiPlane) {
        const bool eq = a == bk;
        const int32_t dg = hd + (eq ? MATCH_SCORE : MISMATCH_SCORE);
        const int32_t vg = hv + GAP_SCORE;                          // from (i, j-1): diagonal b+1
        const int32_t hg = hh + GAP_SCORE;                          // from (i-1, j): diagonal b-1
        const bool isV = vg > dg;
        const int32_t m1 = max(dg, vg);
        const bool isH = hg > m1;
        int32_t v = max(m1, hg);
        const bool valid = uint32_t(s - lo[c]) <= span[c];
        v = (s == lo[c]) ? 0 : v;                                   // i == 0 or j == 0: free leading gaps
        H[c] = valid ? v : H[c];
        loPlane = __ballot(isH || (!isV && !eq));
        hiPlane = __ballot(isV || isH);
    };

    int32_t s = geo.s0;
    for(uint32_t it = 0; it < iters; it++, s += 2) {
        uint64_t words[RW];
        {   // anti-diagonal s: even c hold cells
            int32_t left = __shfl_up(H[C - 1], 1, G); if(l == 0) left = NEG_SCORE;
#pragma unroll
            for(int c = 0; c < C; c += 2) {
                const int32_t hh = (c == 0) ? left : H[c == 0 ? 0 : c - 1];
                cell(c, s, aw[c / 2], bw[c / 2], H[c], H[c + 1], hh, words[2 * c], words[2 * c + 1]);
            }
        }
        {   // anti-diagonal s+1: odd c hold cells
            int32_t right = __shfl_down(H[0], 1, G); if(l == G - 1) right = NEG_SCORE;
#pragma unroll
            for(int c = 1; c < C; c += 2) {
                const int32_t hv = (c == C - 1) ? right : H[c == C - 1 ? c : c + 1];
                cell(c, s + 1, aw[c / 2 + 1], bw[c / 2], H[c], hv, H[c - 1], words[2 * c], words[2 * c + 1]);
            }
        }
        // Lane k stores word k of this iteration's trace record.
        uint64_t mine = words[0];
#pragma unroll
        for(int k = 1; k < RW; k++) mine = (lane == k) ? words[k] : mine;
        if(lane < RW) tr[uint64_t(it) * RW + lane] = mine;
        // Slide the windows.
#pragma unroll
        for(int k = 0; k < HC; k++) aw[k] = aw[k + 1];
        aw[HC] = aNext1; aNext1 = aNext2;
#pragma unroll
        for(int h = HC - 1; h >= 1; h--) bw[h] = bw[h - 1];
        bw[0] = bNext1; bNext1 = bNext2;
        ++ib;
        aNext2 = loadA(ib + 1 + l * HC + HC);
        bNext2 = loadB(ib + 1 - bandMin - l * HC);
    }

    // End cell: maximum over the border cells = final value of every diagonal; ties to the smallest (i, j).
    int32_t bestScore = NEG_SCORE, bestI = 0x7fffffff, bestJ = 0x7fffffff;
#pragma unroll
    for(int c = 0; c < C; c++) {
        const int32_t d = bandMin + l * C + c;
        const int32_t i = (d >= nx - ny) ? nx : ny + d, j = i - d;
        const int32_t v = (lo[c] != 0x40000000) ? H[c] : NEG_SCORE;
        if(v > bestScore || (v == bestScore && v > NEG_SCORE && (i < bestI || (i == bestI && j < bestJ)))) { bestScore = v; bestI = i; bestJ = j; }
    }
#pragma unroll
    for(int d = G / 2; d >= 1; d >>= 1) {
        const int32_t os = __shfl_xor(bestScore, d, G);
        const int32_t oi = __shfl_xor(bestI, d, G);
        const int32_t oj = __shfl_xor(bestJ, d, G);
        if(os > bestScore || (os == bestScore && (oi < bestI || (oi == bestI && oj < bestJ)))) { bestScore = os; bestI = oi; bestJ = oj; }
    }
    if(hasTask && l == 0) {
        DpEnd e; e.traceOffset = bundleOffsets[bundle]; e.bestI = bestI; e.bestJ = bestJ; e.score = bestScore; e.laneBase = uint32_t(g * G);
        ends[t] = e;
    }
}

// ---- forward kernel, second version ---------------------------------------------------------
// Same tasks, bundles, trace format and DpEnd as bandedDpForwardKernel, which stays beside it
// (SHASTA_MI355X_DP_FORWARD=1) until this one has been timed on the MI355X.  Every change comes
// from the first version's ISA (75 VALU instructions per iteration for two cells per lane,
// scripts/isa_loop.py):
//  * three phases, general / steady / general.  In the steady phase (all but about a band width
//    of iterations at either end) every cell of the wavefront that exists is inside the matrix and
//    past the first cell of its diagonal, and every kmer-id load is in range: no validity tests,
//    no index clamps.  It runs in blocks of DP_BLOCK iterations, fully unrolled: a lane's kmer ids
//    of a block are DP_BLOCK + C/2 consecutive elements per read, fetched as one 16-byte load per
//    read and block, one block ahead -- no sliding register windows, no per-iteration address;
//  * scores are kept biased by -NEG_SCORE, so "outside the band" is 0 and the neighbour exchange
//    is one DPP shift with zero fill (row_shr/shl for 16-lane groups, wave_shr/shl otherwise)
//    instead of ds_bpermute + select -- same decisions: max, compare and adding a constant commute
//    with the bias, and nothing overflows (|score| < 2^27);
//  * trace planes: one ballot per comparison (the mask v_cmp wrote anyway), combined on the
//    scalar unit; the ballot of a combined predicate is compiled to v_cndmask + v_cmp;
//  * the trace record goes from lane 0 into a 256-byte LDS line per wavefront and leaves as one
//    coalesced 4-byte store per lane when the line is full, instead of a select chain over the
//    lanes and a partial store every iteration;
//  * trip counts are made scalar (readfirstlane), so loop control runs on the scalar unit.
constexpr int DP_BLOCK = 4;
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// The value of the lane below / above in a G-lane group; 0 at the group's edge.
template<int G> __device__ __forceinline__ int32_t fromLaneBelow(int32_t v, int l)
{
    constexpr int ctrl = (G == 16) ? 0x111 : 0x138;             // row_shr:1 : wave_shr:1; bound_ctrl = zero fill
    int32_t r = __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true);
    if constexpr (G == 32) r = (l == 0) ? 0 : r;
    return r;
}
template<int G> __device__ __forceinline__ int32_t fromLaneAbove(int32_t v, int l)
{
    constexpr int ctrl = (G == 16) ? 0x101 : 0x130;             // row_shl:1 : wave_shl:1
    int32_t r = __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true);
    if constexpr (G == 32) r = (l == G - 1) ? 0 : r;
    return r;
}
struct __attribute__((packed, aligned(4))) KmerQuad { uint32_t v[4]; };     // four consecutive kmer ids, 4-byte aligned

template<int G, int C>
__global__ void __launch_bounds__(256)
bandedDpForwardKernel2(
    const uint32_t* __restrict__ kmerIds, const PairDesc* __restrict__ pairs, const DpTask* __restrict__ tasks,
    const uint32_t* __restrict__ sortedIds, uint32_t taskCount,
    const uint64_t* __restrict__ bundleOffsets, uint32_t bundleCount,
    uint64_t* __restrict__ trace, DpEnd* __restrict__ ends)
{
    constexpr int T = WAVE / G, HC = C / 2, RW = 2 * C, U = DP_BLOCK;
    constexpr int F = 32 / RW;                            // iterations per 256-byte trace line
    constexpr int AL = F > U ? F : U;                     // steady iterations come in groups of AL: whole blocks, whole lines
    constexpr int32_t BIAS = -NEG_SCORE, NO_DIAGONAL = 0x40000000;
    static_assert(C >= 2 && C <= 16 && U >= 3 && AL % U == 0 && AL % F == 0, "block / line geometry");
    __shared__ __attribute__((aligned(16))) uint64_t traceLines[4 * 32];   // one 256-byte line per wavefront of the block (16-byte LDS writes)
    const int lane = laneId();
    const uint32_t bundle = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if(bundle >= bundleCount) return;                     // whole wave leaves: all 64 lanes are active below, no block barriers
    uint64_t* const line = traceLines + 32 * (threadIdx.x >> 6);
    const int g = lane / G, l = lane % G;
    const uint32_t pos = bundle * T + uint32_t(g);
    const bool hasTask = pos < taskCount;
    const uint32_t t = sortedIds[hasTask ? pos : bundle * T];
    const DpTask task = tasks[t];
    const PairDesc pd = pairs[task.pair];
    const uint32_t* __restrict__ p0 = kmerIds + pd.begin0;
    const uint32_t* __restrict__ p1 = kmerIds + pd.begin1;
    const int32_t nx = int32_t(pd.nx), ny = int32_t(pd.ny);
    const int32_t bandMin = task.bandMin, width = task.bandMax - task.bandMin + 1;
    const DpGeometry geo = dpGeometry(task.bandMin, task.bandMax, pd.nx, pd.ny);
    uint32_t itersLane = geo.iters;
#pragma unroll
    for(int d = G; d < WAVE; d <<= 1) itersLane = max(itersLane, uint32_t(__shfl_xor(int(itersLane), d, WAVE)));
    const uint32_t iters = __builtin_amdgcn_readfirstlane(itersLane);
    uint64_t* __restrict__ tr = trace + bundleOffsets[bundle];

    // Per diagonal: first and last anti-diagonal that hold a cell of the matrix.
    int32_t lo[C];
    uint32_t span[C];
    bool exists[C];
#pragma unroll
    for(int c = 0; c < C; c++) {
        const int32_t b = l * C + c, d = bandMin + b;
        const int32_t first = d < 0 ? -d : d;
        const int32_t last = min(2 * nx - d, 2 * ny + d);
        exists[c] = hasTask && b < width && d <= nx && d >= -ny && last >= first;
        lo[c] = exists[c] ? first : NO_DIAGONAL;
        span[c] = exists[c] ? uint32_t(last - first) : 0u;
    }
    int32_t H[C];                                         // biased: score + BIAS; 0 = no cell
#pragma unroll
    for(int c = 0; c < C; c++) H[c] = 0;

    // Iteration `it` works at ib = ib0 + it: it compares A[ib + l HC - 1 + k], k = 0..HC, with
    // B[ib - bandMin - l HC - 1 - h], h = 0..HC-1.
    const int32_t ib0 = (geo.s0 + bandMin) / 2;
    auto loadA = [&](int32_t idx) { return p0[min(max(idx, 0), nx - 1)]; };
    auto loadB = [&](int32_t idx) { return p1[min(max(idx, 0), ny - 1)]; };

    // One anti-diagonal pair.  STEADY: every existing cell is valid and past its first cell.
    auto cell = [&](auto steadyTag, int c, int32_t sc, uint32_t a, uint32_t bk, int32_t hd, int32_t hv, int32_t hh, uint64_t& loPlane, uint64_t& hiPlane) {
        constexpr bool STEADY = decltype(steadyTag)::value;
        const bool eq = a == bk;
        const int32_t dg = hd + (eq ? MATCH_SCORE - GAP_SCORE : MISMATCH_SCORE - GAP_SCORE);   // the three candidates before the gap penalty they share
        const bool isV = hv > dg;                                   // from (i, j-1): diagonal b+1
        const int32_t m1 = max(dg, hv);
        const bool isH = hh > m1;                                   // from (i-1, j): diagonal b-1
        int32_t v = max(m1, hh) + GAP_SCORE;
        if constexpr (STEADY) {
            SHASTA_DEVICE_CHECK(!exists[c] || (sc > lo[c] && uint32_t(sc - lo[c]) <= span[c]));
            H[c] = exists[c] ? v : 0;
        } else {
            const bool valid = uint32_t(sc - lo[c]) <= span[c];
            v = (sc == lo[c]) ? BIAS : v;                           // i == 0 or j == 0: free leading gaps
            H[c] = valid ? v : H[c];
        }
        const uint64_t bEq = ballot64(eq), bV = ballot64(isV), bH = ballot64(isH);
        loPlane = bH | ~(bV | bEq);                                 // codes: 0 diagonal+equal, 1 diagonal+different, 2 vertical, 3 horizontal
        hiPlane = bV | bH;
    };
    // aw(k), bw(h): the kmer ids of this iteration.
    auto antiDiagonals = [&](auto steadyTag, int32_t s, auto aw, auto bw, uint64_t (&words)[RW]) {
        {   // anti-diagonal s: even c hold cells
            const int32_t left = fromLaneBelow<G>(H[C - 1], l);
#pragma unroll
            for(int c = 0; c < C; c += 2) {
                const int32_t hh = (c == 0) ? left : H[c == 0 ? 0 : c - 1];
                cell(steadyTag, c, s, aw(c / 2), bw(c / 2), H[c], H[c + 1], hh, words[2 * c], words[2 * c + 1]);
            }
        }
        {   // anti-diagonal s+1: odd c hold cells
            const int32_t right = fromLaneAbove<G>(H[0], l);
#pragma unroll
            for(int c = 1; c < C; c += 2) {
                const int32_t hv = (c == C - 1) ? right : H[c == C - 1 ? c : c + 1];
                cell(steadyTag, c, s + 1, aw(c / 2 + 1), bw(c / 2), H[c], hv, H[c - 1], words[2 * c], words[2 * c + 1]);
            }
        }
    };
    // The record of an iteration goes to slot (it mod F) of the wavefront's line; a full line leaves as
    // one coalesced store.  Lane 0 writes, all lanes read: the wave barrier keeps the compiler from
    // moving the read up (the hardware runs a wavefront's LDS operations in order).
    auto putRecord = [&](int slot, const uint64_t (&words)[RW]) {
        if(lane == 0) {
            ulonglong2* __restrict__ record = reinterpret_cast<ulonglong2*>(line + slot * RW);
#pragma unroll
            for(int k = 0; k < C; k++) { ulonglong2 w; w.x = words[2 * k]; w.y = words[2 * k + 1]; record[k] = w; }
        }
    };
    auto flushLine = [&](uint32_t lineIndex) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        SHASTA_DEVICE_CHECK(uint64_t(lineIndex) * 32 + 32 <= ((uint64_t(iters) * RW + 31) & ~31ULL));      // inside the bundle's trace (dpBundleKernel)
        const uint32_t d = reinterpret_cast<const uint32_t*>(line)[lane];
        reinterpret_cast<uint32_t*>(tr + uint64_t(lineIndex) * 32)[lane] = d;
        __builtin_amdgcn_wave_barrier();
    };

    // General iterations [from, to): sliding register windows fed by clamped loads two iterations ahead.
    auto general = [&](uint32_t from, uint32_t to) {
        if(from >= to) return;
        const int32_t ib = ib0 + int32_t(from);
        uint32_t aw[HC + 1], bw[HC];
#pragma unroll
        for(int k = 0; k <= HC; k++) aw[k] = loadA(ib + l * HC - 1 + k);
#pragma unroll
        for(int h = 0; h < HC; h++) bw[h] = loadB(ib - bandMin - l * HC - 1 - h);
        uint32_t aNext1 = loadA(ib + l * HC + HC), aNext2 = loadA(ib + 1 + l * HC + HC);
        uint32_t bNext1 = loadB(ib - bandMin - l * HC), bNext2 = loadB(ib + 1 - bandMin - l * HC);
        for(uint32_t it = from; it < to; it++) {
            uint64_t words[RW];
            antiDiagonals(std::false_type{}, geo.s0 + 2 * int32_t(it), [&](int k) { return aw[k]; }, [&](int h) { return bw[h]; }, words);
            putRecord(int(it % F), words);
            if(it % F == F - 1) flushLine(it / F);
#pragma unroll
            for(int k = 0; k < HC; k++) aw[k] = aw[k + 1];
            aw[HC] = aNext1; aNext1 = aNext2;
#pragma unroll
            for(int h = HC - 1; h >= 1; h--) bw[h] = bw[h - 1];
            bw[0] = bNext1; bNext1 = bNext2;
            aNext2 = loadA(ib0 + int32_t(it) + 2 + l * HC + HC);
            bNext2 = loadB(ib0 + int32_t(it) + 2 - bandMin - l * HC);
        }
    };

    // Steady iterations.  A cell (lane, c) is steady at `it` when lo < s0 + 2 it + (c & 1) <= lo + span;
    // the block that starts at itB loads A[iaBlock + itB + j], B[jbBlock + itB + j], j = 0..U-1
    // (the new elements of the block after it).
    const int32_t iaBlock = ib0 + U + l * HC + HC - 1, jbBlock = ib0 + U - bandMin - l * HC - 1;
    int32_t itLo = 0, itHi = int32_t(iters) - 1;          // cells steady on [itLo, itHi]
    int32_t startLo = max(-iaBlock, -jbBlock), startHi = min(nx - U - iaBlock, ny - U - jbBlock);   // block starts whose loads are in range
#pragma unroll
    for(int c = 0; c < C; c++) {
        if(exists[c]) {
            itLo = max(itLo, (lo[c] + 2 - geo.s0 - (c & 1)) >> 1);
            itHi = min(itHi, (lo[c] + int32_t(span[c]) - geo.s0 - (c & 1)) >> 1);
        }
    }
#pragma unroll
    for(int d = 1; d < WAVE; d <<= 1) {
        itLo = max(itLo, __shfl_xor(itLo, d, WAVE)); itHi = min(itHi, __shfl_xor(itHi, d, WAVE));
        startLo = max(startLo, __shfl_xor(startLo, d, WAVE)); startHi = min(startHi, __shfl_xor(startHi, d, WAVE));
    }
    // Groups of AL iterations starting at multiples of AL: the first at steadyBegin, every block start in
    // [startLo, startHi], every iteration in [itLo, itHi].
    const int32_t firstStart = (max(max(itLo, startLo), 0) + AL - 1) / AL * AL;
    const int32_t lastGroupStart = min(itHi - (AL - 1), startHi - (AL - U));
    const uint32_t groups = __builtin_amdgcn_readfirstlane(uint32_t(lastGroupStart >= firstStart ? (lastGroupStart - firstStart) / AL + 1 : 0));
    const uint32_t steadyBegin = __builtin_amdgcn_readfirstlane(uint32_t(firstStart));

    if(groups == 0) {
        general(0, iters);
    } else {
        general(0, steadyBegin);
        {
            // Block registers: a[x] = A[ib + l HC - 1 + x], x = 0..U+HC-1; e[x] = B[ib - bandMin - l HC - HC + x], x = 0..U+HC-2.
            // Iteration u of the block: aw(k) = a[u + k], bw(h) = e[u + HC - 1 - h].
            const int32_t ib = ib0 + int32_t(steadyBegin);
            uint32_t a[U + HC], e[U + HC - 1];
#pragma unroll
            for(int x = 0; x < U + HC; x++) a[x] = loadA(ib + l * HC - 1 + x);
#pragma unroll
            for(int x = 0; x < U + HC - 1; x++) e[x] = loadB(ib - bandMin - l * HC - HC + x);
            const uint32_t* __restrict__ pa = p0 + (int64_t(iaBlock) + int64_t(steadyBegin));
            const uint32_t* __restrict__ pb = p1 + (int64_t(jbBlock) + int64_t(steadyBegin));
            uint32_t lineIndex = steadyBegin / F;
            for(uint32_t grp = 0; grp < groups; grp++) {
#pragma unroll
                for(int blk = 0; blk < AL / U; blk++) {
                    SHASTA_DEVICE_CHECK(pa >= p0 && pa + U <= p0 + nx && pb >= p1 && pb + U <= p1 + ny);
                    const KmerQuad newA = *reinterpret_cast<const KmerQuad*>(pa);
                    const KmerQuad newB = *reinterpret_cast<const KmerQuad*>(pb);
                    pa += U; pb += U;
#pragma unroll
                    for(int u = 0; u < U; u++) {
                        uint64_t words[RW];
                        antiDiagonals(std::true_type{}, geo.s0 + 2 * int32_t(steadyBegin + grp * AL + blk * U + u), [&](int k) { return a[u + k]; }, [&](int h) { return e[u + HC - 1 - h]; }, words);
                        const int slot = (blk * U + u) % F;
                        putRecord(slot, words);
                        if(slot == F - 1) { flushLine(lineIndex); ++lineIndex; }
                    }
#pragma unroll
                    for(int x = 0; x < HC; x++) a[x] = a[x + U];
#pragma unroll
                    for(int j = 0; j < U; j++) a[HC + j] = newA.v[j];
#pragma unroll
                    for(int x = 0; x < HC - 1; x++) e[x] = e[x + U];
#pragma unroll
                    for(int j = 0; j < U; j++) e[HC - 1 + j] = newB.v[j];
                }
            }
        }
        general(steadyBegin + groups * AL, iters);
    }
    if(iters % F != 0) flushLine(iters / F);              // the last, partial line (the bundle's trace is a whole number of lines)

    // End cell: maximum over the border cells = final value of every diagonal; ties to the smallest (i, j).
    int32_t bestScore = NEG_SCORE, bestI = 0x7fffffff, bestJ = 0x7fffffff;
#pragma unroll
    for(int c = 0; c < C; c++) {
        const int32_t d = bandMin + l * C + c;
        const int32_t i = (d >= nx - ny) ? nx : ny + d, j = i - d;
        const int32_t v = exists[c] ? H[c] - BIAS : NEG_SCORE;
        if(v > bestScore || (v == bestScore && v > NEG_SCORE && (i < bestI || (i == bestI && j < bestJ)))) { bestScore = v; bestI = i; bestJ = j; }
    }
#pragma unroll
    for(int d = G / 2; d >= 1; d >>= 1) {
        const int32_t os = __shfl_xor(bestScore, d, G);
        const int32_t oi = __shfl_xor(bestI, d, G);
        const int32_t oj = __shfl_xor(bestJ, d, G);
        if(os > bestScore || (os == bestScore && (oi < bestI || (oi == bestI && oj < bestJ)))) { bestScore = os; bestI = oi; bestJ = oj; }
    }
    if(hasTask && l == 0) {
        DpEnd e; e.traceOffset = bundleOffsets[bundle]; e.bestI = bestI; e.bestJ = bestJ; e.score = bestScore; e.laneBase = uint32_t(g * G);
        ends[t] = e;
    }
}

// One lane per task: walk the path from the end cell through the packed trace.  The trace is
// consumed in chunks of CW words (128 or 256 bytes, whole cache lines): the chunk under the
// path sits in the lane's private LDS window, the next one (the path only moves towards smaller
// anti-diagonals) is already in flight in registers, so every line is fetched once and its
// latency is covered by the walk through the previous chunk.
template<int CW>
__global__ void __launch_bounds__(256)
dpTracebackKernel(
    const PairDesc* __restrict__ pairs, const DpTask* __restrict__ tasks, const uint32_t* __restrict__ sortedIds, uint32_t taskCount,
    const DpEnd* __restrict__ ends, const uint64_t* __restrict__ trace,
    const uint64_t* __restrict__ ordOffsets, uint32_t* __restrict__ ordScratch,
    DpResult* __restrict__ results, DeviceOptions opt, unsigned long long* __restrict__ pairBest)
{
    constexpr int QUADS = CW / 2;                          // 16-byte pieces of a chunk
    __shared__ uint4 window[256 * QUADS];                  // [piece][thread]: conflict-free for a wave
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if(idx >= taskCount) return;
    const uint32_t t = sortedIds[idx];
    const DpTask task = tasks[t];
    const PairDesc pd = pairs[task.pair];
    const DpEnd e = ends[t];
    const DpGeometry geo = dpGeometry(task.bandMin, task.bandMax, pd.nx, pd.ny);
    const int C = dpDiagonals(geo.cls);
    const uint32_t RW = uint32_t(2 * C);
    const uint32_t itersPerChunk = uint32_t(CW) / RW;
    const uint4* __restrict__ tr = reinterpret_cast<const uint4*>(trace + e.traceOffset);
    const uint64_t ordBase = ordOffsets[t];
    uint32_t pos = min(pd.nx, pd.ny);
    uint32_t count = 0, prevX = 0, prevY = 0, last0 = 0, last1 = 0, first0 = 0, first1 = 0, maxSkip = 0, maxDrift = 0;
    int32_t minOffset = 0x7fffffff, maxOffset = int32_t(0x80000000);
    long long sumOffset = 0;
    int32_t i = e.bestI, j = e.bestJ;
    const bool ok = e.score > NEG_SCORE;
    // Epochs: every lane moves its prefetched chunk into the window and prefetches the next one at
    // the same point of the program, then walks until its path leaves the chunk.  The wave waits
    // for memory once per epoch, for loads issued a whole epoch earlier.
    bool active = ok && i > 0 && j > 0;
    int64_t chunk = active ? int64_t((uint32_t(i + j - geo.s0) >> 1) / itersPerChunk) : -1;
    uint4 next[QUADS];
#pragma unroll
    for(int k = 0; k < QUADS; k++) next[k] = active ? tr[chunk * QUADS + k] : make_uint4(0, 0, 0, 0);
    while(__any(active)) {
        if(active) {
#pragma unroll
            for(int k = 0; k < QUADS; k++) window[k * 256 + threadIdx.x] = next[k];
            if(chunk > 0) {
#pragma unroll
                for(int k = 0; k < QUADS; k++) next[k] = tr[(chunk - 1) * QUADS + k];
            }
        }
        while(active) {
            const int32_t b = i - j - task.bandMin;
            const uint32_t it = uint32_t(i + j - geo.s0) >> 1;
            if(int64_t(it / itersPerChunk) != chunk) break;
            const uint32_t c = uint32_t(b) % uint32_t(C), bit = e.laneBase + uint32_t(b) / uint32_t(C);
            const uint32_t word = (it % itersPerChunk) * RW + 2 * c;           // even: one 16-byte piece
            const uint4 w = window[(word >> 1) * 256 + threadIdx.x];
            const uint64_t lo = uint64_t(w.x) | (uint64_t(w.y) << 32), hi = uint64_t(w.z) | (uint64_t(w.w) << 32);
            const uint32_t dir = uint32_t((lo >> bit) & 1ULL) | (uint32_t((hi >> bit) & 1ULL) << 1);
            if(dir == 0u) {
                // A diagonal step over equal kmers: an aligned marker pair (src/Align4.cpp:1057-1061).
                const uint32_t x = uint32_t(i - 1), y = uint32_t(j - 1);
                --pos;
                *reinterpret_cast<uint2*>(ordScratch + 2 * (ordBase + pos)) = make_uint2(x, y);
                const int32_t offset = int32_t(x) - int32_t(y);
                if(count == 0) { last0 = x; last1 = y; }
                else {
                    maxSkip = max(maxSkip, max(prevX - x, prevY - y));
                    const int32_t prevOffset = int32_t(prevX) - int32_t(prevY);
                    const int32_t drift = offset - prevOffset;
                    maxDrift = max(maxDrift, uint32_t(drift < 0 ? -drift : drift));
                }
                minOffset = min(minOffset, offset); maxOffset = max(maxOffset, offset);
                sumOffset += offset;
                first0 = x; first1 = y; prevX = x; prevY = y;
                ++count;
                --i; --j;
            } else if(dir == 1u) { --i; --j; }
            else if(dir == 2u) { --j; }
            else { --i; }
            active = i > 0 && j > 0;
        }
        --chunk;
    }
    DpResult r;
    r.ordBegin = ordBase + pos;
    r.sumOffset = sumOffset;
    r.markerCount = count; r.first0 = first0; r.first1 = first1; r.last0 = last0; r.last1 = last1;
    r.minOffset = minOffset; r.maxOffset = maxOffset; r.maxSkip = maxSkip; r.maxDrift = maxDrift;
    r.score = e.score; r.pad = 0;
    // Inner acceptance, src/Align4.cpp:944-981.
    bool pass = count > 0 && uint64_t(count) >= opt.minAlignedMarkerCount;
    if(pass) {
        const double f0 = double(count) / double(last0 + 1 - first0);
        const double f1 = double(count) / double(last1 + 1 - first1);
        if(min(f0, f1) < opt.minAlignedFraction) pass = false;
        if(uint64_t(maxSkip) > opt.maxSkip || uint64_t(maxDrift) > opt.maxDrift) pass = false;
        const uint32_t leftTrim = min(first0, first1);
        const uint32_t rightTrim = min(pd.nx - 1 - last0, pd.ny - 1 - last1);
        if(uint64_t(leftTrim) > opt.maxTrim || uint64_t(rightTrim) > opt.maxTrim) pass = false;
    }
    r.passes = pass ? 1u : 0u;
    results[t] = r;
    // Best component = most aligned markers (:132-139); ties resolved towards the
    // component whose first cell in (iY,iX) order comes first, and flagged later.
    if(pass) atomicMax(&pairBest[task.pair], ((unsigned long long)count << 32) | (unsigned long long)(0xffffffffu - task.label));
}

__global__ void __launch_bounds__(256)
winnerKernel(const DpTask* __restrict__ tasks, const DpResult* __restrict__ results, uint32_t taskCount,
    const unsigned long long* __restrict__ pairBest, uint32_t* __restrict__ pairWinner, uint8_t* __restrict__ pairTie)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= taskCount) return;
    const DpResult r = results[t];
    if(!r.passes) return;
    const DpTask task = tasks[t];
    const unsigned long long best = pairBest[task.pair];
    const unsigned long long key = ((unsigned long long)r.markerCount << 32) | (unsigned long long)(0xffffffffu - task.label);
    if(key == best) pairWinner[task.pair] = t;
    else if((key >> 32) == (best >> 32)) pairTie[task.pair] = 1;
}

// Per candidate: AlignmentInfo (src/Alignment.cpp:67-113) and the outer filters of
// src/AssemblerAlign.cpp:439-472.
__global__ void __launch_bounds__(256)
finalizeKernel(const PairDesc* __restrict__ pairs, const shasta_oriented_read_pair* __restrict__ candidates, uint32_t pairCount,
    const DpResult* __restrict__ results, const unsigned long long* __restrict__ pairBest,
    const uint32_t* __restrict__ pairWinner, const uint8_t* __restrict__ pairTie, const uint8_t* __restrict__ pairFlags,
    DeviceOptions opt, int wantOrdinals,
    uint8_t* __restrict__ status, shasta_alignment_data* __restrict__ rows,
    uint32_t* __restrict__ storedFlags, uint64_t* __restrict__ ordCounts)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if(p == pairCount) { storedFlags[p] = 0; ordCounts[p] = 0; return; }
    if(p > pairCount) return;
    uint8_t st;
    uint32_t stored = 0;
    uint64_t ordCount = 0;
    if(pairFlags[p]) {
        st = SHASTA_ALIGN_SKIPPED;
    } else if(pairBest[p] == 0) {
        st = SHASTA_ALIGN_EMPTY;
    } else {
        const DpResult r = results[pairWinner[p]];
        const PairDesc pd = pairs[p];
        shasta_alignment_data row;
        row.pair = candidates[p];
        row.pair.isSameStrand = row.pair.isSameStrand ? 1 : 0;
        row.pair.pad[0] = row.pair.pad[1] = row.pair.pad[2] = 0;
        row.info.data[0].markerCount = pd.nx; row.info.data[0].firstOrdinal = r.first0; row.info.data[0].lastOrdinal = r.last0;
        row.info.data[1].markerCount = pd.ny; row.info.data[1].firstOrdinal = r.first1; row.info.data[1].lastOrdinal = r.last1;
        row.info.markerCount = r.markerCount;
        row.info.minOrdinalOffset = r.minOffset; row.info.maxOrdinalOffset = r.maxOffset;
        row.info.averageOrdinalOffset = int32_t(round(double(r.sumOffset) / double(r.markerCount)));
        row.info.maxSkip = r.maxSkip; row.info.maxDrift = r.maxDrift;
        row.info.isInReadGraph = 0; row.info.pad[0] = row.info.pad[1] = row.info.pad[2] = 0;
        rows[p] = row;
        bool good = uint64_t(r.markerCount) >= opt.minAlignedMarkerCount;
        const double f0 = double(r.markerCount) / double(r.last0 + 1 - r.first0);
        const double f1 = double(r.markerCount) / double(r.last1 + 1 - r.first1);
        if(min(f0, f1) < opt.minAlignedFraction) good = false;
        const uint32_t lt0 = r.first0, lt1 = r.first1, rt0 = pd.nx - 1 - r.last0, rt1 = pd.ny - 1 - r.last1;
        if(uint64_t(min(lt0, lt1)) > opt.maxTrim || uint64_t(min(rt0, rt1)) > opt.maxTrim) good = false;
        if(uint64_t(r.maxSkip) > opt.maxSkip || uint64_t(r.maxDrift) > opt.maxDrift) good = false;
        if(opt.suppressContainments) {
            const uint32_t mt = uint32_t(opt.maxTrim);
            if((lt0 <= mt && rt0 <= mt) || (lt1 <= mt && rt1 <= mt)) good = false;       // isContaining
        }
        st = good ? SHASTA_ALIGN_STORED : SHASTA_ALIGN_REJECTED;
        if(pairTie[p]) st |= SHASTA_ALIGN_TIE_FLAG;
        stored = good ? 1u : 0u;
        ordCount = (wantOrdinals || good) ? r.markerCount : 0;
    }
    status[p] = st;
    storedFlags[p] = stored;
    ordCounts[p] = ordCount;
}

__global__ void __launch_bounds__(256)
gatherOrdinalsKernel(const DpResult* __restrict__ results, const uint32_t* __restrict__ pairWinner,
    const uint64_t* __restrict__ ordToc, uint32_t pairCount, const uint32_t* __restrict__ ordScratch, uint32_t* __restrict__ ordOut)
{
    // One wave per candidate copies its alignment.
    const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if(p >= pairCount) return;
    const uint64_t begin = ordToc[p], n = ordToc[p + 1] - begin;
    if(n == 0) return;
    const uint64_t src = results[pairWinner[p]].ordBegin;
    for(uint64_t k = laneId(); k < 2 * n; k += WAVE) ordOut[2 * begin + k] = ordScratch[2 * src + k];
}

// shasta::compress (src/compressAlignment.cpp:11-67; formats compressAlignment.hpp:101-321).
// A streak is a maximal run of marker pairs that advance both ordinals by one; its record holds
// (skip0, skip1) from the last pair of the previous streak (from (0,0) for the first) and its
// length, in the smallest of five formats (1/2/4/8/16 bytes).
struct StreakRecord { uint64_t bits; uint32_t w[3]; int len; };

__device__ __forceinline__ StreakRecord makeStreakRecord(int32_t skip0, int32_t skip1, uint32_t streak)
{
    StreakRecord r;
    const uint64_t u0 = uint32_t(skip0), u1 = uint32_t(skip1), nm1 = uint64_t(streak) - 1;
    r.w[0] = uint32_t(skip0); r.w[1] = uint32_t(skip1); r.w[2] = uint32_t(nm1);
    if(skip0 >= 0 && skip0 <= 3 && skip1 >= 0 && skip1 <= 3 && streak <= 8) {
        r.bits = 0 | (u0 & 3) << 1 | (u1 & 3) << 3 | (nm1 & 7) << 5; r.len = 1;
    } else if(skip0 >= -8 && skip0 <= 7 && skip1 >= -8 && skip1 <= 7 && streak <= 32) {
        r.bits = 1 | (u0 & 0xf) << 3 | (u1 & 0xf) << 7 | (nm1 & 0x1f) << 11; r.len = 2;
    } else if(skip0 >= -512 && skip0 <= 511 && skip1 >= -512 && skip1 <= 511 && streak <= 512) {
        r.bits = 3 | (u0 & 0x3ff) << 3 | (u1 & 0x3ff) << 13 | (nm1 & 0x1ff) << 23; r.len = 4;
    } else if(skip0 >= -524288 && skip0 <= 524287 && skip1 >= -524288 && skip1 <= 524287 && streak <= 2097152) {
        r.bits = 5 | (u0 & 0xfffff) << 3 | (u1 & 0xfffff) << 23 | (nm1 & 0x1fffff) << 43; r.len = 8;
    } else {
        r.bits = 7; r.len = 16;
    }
    return r;
}

__device__ __forceinline__ void writeStreakRecord(const StreakRecord& r, uint8_t* __restrict__ out)
{
    if(r.len == 16) {
        const uint32_t w[4] = {7u, r.w[0], r.w[1], r.w[2]};
        for(int k = 0; k < 16; k++) out[k] = uint8_t(w[k >> 2] >> (8 * (k & 3)));
    } else {
        for(int k = 0; k < r.len; k++) out[k] = uint8_t(r.bits >> (8 * k));
    }
}

// One wavefront per stored alignment: lanes flag the streak starts of 64 marker pairs at a time;
// a start lane knows its skips at once and its length when the next start is seen (the last
// start of a chunk is carried to the next chunk).  WRITE=false only counts the bytes.
template<bool WRITE>
__device__ __forceinline__ uint64_t compressAlignmentWave(const uint32_t* __restrict__ ord, uint32_t n, uint8_t* __restrict__ out)
{
    const int lane = laneId();
    uint64_t bytes = 0;                       // wave-uniform
    bool havePending = false;                 // wave-uniform: a streak whose end is not known yet
    uint32_t pendingStart = 0; int32_t pendingSkip0 = 0, pendingSkip1 = 0;
    uint32_t carryX = 0, carryY = 0;          // last pair of the previous chunk ((0,0) before the first)
    for(uint32_t base = 0; base < n; base += WAVE) {
        const uint32_t i = base + lane;
        const bool valid = i < n;
        uint2 xy = make_uint2(0, 0);
        if(valid) xy = *reinterpret_cast<const uint2*>(ord + 2 * uint64_t(i));
        uint32_t px = __shfl_up(xy.x, 1, WAVE), py = __shfl_up(xy.y, 1, WAVE);
        if(lane == 0) { px = carryX; py = carryY; }
        const bool start = valid && (i == 0 || xy.x != px + 1 || xy.y != py + 1);
        const uint64_t starts = __ballot(start);
        const int32_t skip0 = int32_t(xy.x) - int32_t(px), skip1 = int32_t(xy.y) - int32_t(py);
        // The first start of this chunk closes the pending streak.
        if(havePending && starts) {
            const uint32_t first = base + uint32_t(__ffsll((unsigned long long)starts) - 1);
            const StreakRecord r = makeStreakRecord(pendingSkip0, pendingSkip1, first - pendingStart);
            if(WRITE && lane == 0) writeStreakRecord(r, out + bytes);
            bytes += uint64_t(r.len);
            havePending = false;
        }
        // Starts of this chunk that are closed by a later start of the same chunk.
        const uint64_t later = (starts >> 1) >> lane;
        const bool closed = start && later != 0;
        uint32_t length = closed ? uint32_t(__ffsll((unsigned long long)later)) : 0u;
        StreakRecord r = makeStreakRecord(skip0, skip1, closed ? length : 1u);
        const uint32_t len = closed ? uint32_t(r.len) : 0u;
        // Exclusive prefix of the record lengths over the wave.
        uint32_t inclusive = len;
#pragma unroll
        for(int d = 1; d < WAVE; d <<= 1) { const uint32_t o = __shfl_up(inclusive, d, WAVE); if(lane >= d) inclusive += o; }
        if(WRITE && closed) writeStreakRecord(r, out + bytes + (inclusive - len));
        bytes += uint64_t(__shfl(inclusive, WAVE - 1, WAVE));
        // The last start of the chunk stays pending.
        if(starts) {
            const int lastLane = 63 - __clzll((unsigned long long)starts);
            havePending = true;
            pendingStart = base + uint32_t(lastLane);
            pendingSkip0 = __shfl(skip0, lastLane, WAVE);
            pendingSkip1 = __shfl(skip1, lastLane, WAVE);
        }
        const int lastValid = int(min(uint32_t(WAVE), n - base)) - 1;
        carryX = __shfl(xy.x, lastValid, WAVE); carryY = __shfl(xy.y, lastValid, WAVE);
    }
    if(havePending) {
        const StreakRecord r = makeStreakRecord(pendingSkip0, pendingSkip1, n - pendingStart);
        if(WRITE && lane == 0) writeStreakRecord(r, out + bytes);
        bytes += uint64_t(r.len);
    }
    return bytes;
}

__global__ void __launch_bounds__(256)
compressSizeKernel(const uint32_t* __restrict__ storedFlags, const DpResult* __restrict__ results,
    const uint32_t* __restrict__ pairWinner, const uint32_t* __restrict__ ordScratch, uint32_t pairCount, uint64_t* __restrict__ sizes)
{
    const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if(p > pairCount) return;
    uint64_t s = 0;
    if(p < pairCount && storedFlags[p]) {
        const DpResult r = results[pairWinner[p]];
        s = compressAlignmentWave<false>(ordScratch + 2 * r.ordBegin, r.markerCount, nullptr);
    }
    if(laneId() == 0) sizes[p] = s;
}

__global__ void __launch_bounds__(256)
compressWriteKernel(const uint32_t* __restrict__ storedFlags, const uint32_t* __restrict__ storedIndex,
    const DpResult* __restrict__ results, const uint32_t* __restrict__ pairWinner, const uint32_t* __restrict__ ordScratch,
    uint32_t pairCount, const uint64_t* __restrict__ byteOffsets, uint8_t* __restrict__ bytes,
    uint64_t* __restrict__ compressedToc, const shasta_alignment_data* __restrict__ rows, shasta_alignment_data* __restrict__ rowsOut)
{
    const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if(p >= pairCount || !storedFlags[p]) return;
    const DpResult r = results[pairWinner[p]];
    (void)compressAlignmentWave<true>(ordScratch + 2 * r.ordBegin, r.markerCount, bytes + byteOffsets[p]);
    const uint32_t k = storedIndex[p];
    if(laneId() == 0) compressedToc[k] = byteOffsets[p];
    // The 64-byte AlignmentData row: one dword per lane.
    if(laneId() < 16) reinterpret_cast<uint32_t*>(rowsOut + k)[laneId()] = reinterpret_cast<const uint32_t*>(rows + p)[laneId()];
}

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
    d.suppressContainments = o.suppressContainments ? 1u : 0u;
    return d;
}

struct BatchScratch {
    DeviceBuffer<PairDesc> pairs;
    DeviceBuffer<shasta_oriented_read_pair> candidates;
    DeviceBuffer<DpTask> tasks;
    DeviceBuffer<uint32_t> counters;            // [0]=taskCount, [1..5]=class counts
    DeviceBuffer<uint8_t> pairFlags, pairTie, status;
    DeviceBuffer<unsigned long long> pairBest, dpCells;
    DeviceBuffer<uint32_t> pairWinner, classLists, storedFlags, storedIndex, scanTemp32;
    DeviceBuffer<uint64_t> traceWords, ordCap, scanTemp64, trace, ordCounts, sizes;
    DeviceBuffer<uint32_t> ordScratch, ordOut;
    DeviceBuffer<DpResult> results;
    DeviceBuffer<shasta_alignment_data> rows, rowsOut;
    DeviceBuffer<uint64_t> compressedToc;
    DeviceBuffer<uint8_t> bytes, bigLog2;
    DeviceBuffer<uint32_t> pairList, bigScratch;
    DeviceBuffer<CellsChunk> chunks;
    DeviceBuffer<uint32_t> dpKeysA, dpKeysB, dpIdsA, dpIdsB;    // tasks sorted by (class, iterations)
    DeviceBuffer<uint64_t> bundleWords;
    DeviceBuffer<DpEnd> ends;
    PinnedBuffer pinRows, pinToc, pinBytes, pinStatus, pinOrdToc, pinOrdinals;   // device-to-host staging
    DeviceBuffer<uint64_t> bigOffsets;
    DeviceBuffer<PairDesc> dsPairs;             // align method 3, step 1: the down-sampled pairs
    DeviceBuffer<DpTask> tasks1;                //                         and their (unbanded) DP tasks
    DeviceBuffer<WideTask> wideTasks;           //                         pairs with more than 1024 diagonals
    DeviceBuffer<WideEnd> wideEnds;
};

// A host worker's stream and sort workspace (two workers pipeline the batches of one call).
// `wide` is a side stream for the few wide-band DP tasks (one wavefront each, latency-bound): they
// overlap the narrow classes instead of occupying the GPU alone.
struct WorkStream { hipStream_t stream; RadixSortWorkspace* sortWs; hipStream_t wide; };

constexpr int CELLS_CLASSES = 3;
constexpr int CELLS_NA_LOG2[CELLS_CLASSES] = {11, 12, 13};     // tabled read below 2048 / 4096 / 8192 markers
constexpr int CELLS_SC_LOG2[CELLS_CLASSES] = {11, 12, 12};
constexpr int CELLS_Q[CELLS_CLASSES] = {2, 4, 4};
constexpr int CELLS_SHARED_WAVES[CELLS_CLASSES] = {4, 6, 4};   // waves of a chunk that share the tabled read
constexpr uint32_t CELLS_CHUNK_MAX[CELLS_CLASSES] = {24, 24, 16};
constexpr uint32_t CELLS_SHARE_MIN = 3;                        // smaller chunks run as one wave

// kind 0: one-wave workgroups; kind 1: CELLS_SHARED_WAVES waves share the table.
template<int Q>
void launchCellsChunksQ(Context& ctx, const WorkStream& ws, BatchScratch& b, int cls, int waves, const CellsChunk* chunks, uint32_t count,
    const DeviceOptions& opt, uint32_t magicX, uint32_t magicY, uint32_t taskCapacity)
{
    const size_t bytes = cellsChunkLdsWords(CELLS_NA_LOG2[cls], CELLS_SC_LOG2[cls], Q, waves) * sizeof(uint32_t);
    static bool attributeSet = false;
    if(!attributeSet) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&align4CellsChunkKernel<Q>),
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        attributeSet = true;
    }
    MI355X_ASSERT(bytes <= 160 * 1024 - 1024);
    hipLaunchKernelGGL(align4CellsChunkKernel<Q>, dim3(count), dim3(WAVE * waves), bytes, ws.stream,
        (const uint32_t*)ctx.kmerIds.data(), (const PairDesc*)b.pairs.data(), chunks, count, (const uint32_t*)b.pairList.data(),
        opt, magicX, magicY, b.tasks.data(), b.counters.data(), taskCapacity, b.pairFlags.data());
    HIP_CHECK(hipGetLastError());
}

void launchCellsChunks(Context& ctx, const WorkStream& ws, BatchScratch& b, int cls, int waves, const CellsChunk* chunks, uint32_t count,
    const DeviceOptions& opt, uint32_t magicX, uint32_t magicY, uint32_t taskCapacity)
{
    if(count == 0) return;
    if(CELLS_Q[cls] == 2) launchCellsChunksQ<2>(ctx, ws, b, cls, waves, chunks, count, opt, magicX, magicY, taskCapacity);
    else launchCellsChunksQ<4>(ctx, ws, b, cls, waves, chunks, count, opt, magicX, magicY, taskCapacity);
}

// What a DP runs on: the kmer-id array its pairs index, the pairs, the tasks.
struct DpInput { const uint32_t* kmerIds; const PairDesc* pairs; const DpTask* tasks; };

// Which forward kernel runs: bandedDpForwardKernel2 unless SHASTA_MI355X_DP_FORWARD=1, or unless it
// disagrees with the first version on this device (dpForwardSelfTest, once per process, loud).
int chooseDpForwardVersion();

template<int G, int C>
void launchDpForward(const DpInput& in, hipStream_t stream, BatchScratch& b, const uint32_t* sortedIds, const DpClassLayout& layout, int cls, int version)
{
    const uint32_t taskCount = layout.taskStart[cls + 1] - layout.taskStart[cls];
    const uint32_t bundleCount = layout.bundleStart[cls + 1] - layout.bundleStart[cls];
    if(taskCount == 0) return;
    hipLaunchKernelGGL((version == 1 ? bandedDpForwardKernel<G, C> : bandedDpForwardKernel2<G, C>), dim3(divUp(bundleCount, 4)), dim3(256), 0, stream,
        in.kmerIds, in.pairs, in.tasks,
        sortedIds + layout.taskStart[cls], taskCount,
        (const uint64_t*)(b.bundleWords.data() + layout.bundleStart[cls]), bundleCount,
        b.trace.data(), b.ends.data());
    HIP_CHECK(hipGetLastError());
}

// K10 for the taskCount tasks in b.tasks (pairs in b.pairs): fills b.results, b.ordScratch and
// b.pairBest.  Returns the number of DP cells (sum of nx * bandWidth); forwardSeconds gets the
// HIP-event time of the forward launches when evA/evB are given.
// Timing events of one batch's DP: start/stop around the forward launch of every class and around the traceback.
struct DpEvents {
    hipEvent_t start[DP_CLASSES + 1], stop[DP_CLASSES + 1], fork, join;
    void create() { for(int k = 0; k <= DP_CLASSES; k++) { HIP_CHECK(hipEventCreate(&start[k])); HIP_CHECK(hipEventCreate(&stop[k])); } HIP_CHECK(hipEventCreate(&fork)); HIP_CHECK(hipEventCreate(&join)); }
    void destroy() { for(int k = 0; k <= DP_CLASSES; k++) { (void)hipEventDestroy(start[k]); (void)hipEventDestroy(stop[k]); } (void)hipEventDestroy(fork); (void)hipEventDestroy(join); }
};
struct DpBatchStats { uint64_t cells[DP_CLASSES] = {0}, bytes[DP_CLASSES] = {0}; uint32_t tasks[DP_CLASSES] = {0}; };

// Forward half of K10 for taskCount tasks: sort by (band class, iterations), bundle, lay out the
// trace, run the forward kernel of every class.  Leaves b.trace / b.ends for a traceback kernel.
struct DpForwardState {
    const uint32_t* sortedIds;
    uint32_t classCounts[DP_CLASSES];
    unsigned long long sums[16];          // [0] DP cells, [1] trace word bound, [2+c] cells of class c, [8+c] bytes of class c
};

DpForwardState runDpForward(const WorkStream& ws, BatchScratch& b, const DpInput& in, uint32_t taskCount, bool reserveOrdinals, DpEvents* ev)
{
    hipStream_t stream = ws.stream;
    DpForwardState f;
    const int version = chooseDpForwardVersion();
    b.dpKeysA.reserve(taskCount, stream); b.dpKeysB.reserve(taskCount, stream);
    b.dpIdsA.reserve(taskCount, stream); b.dpIdsB.reserve(taskCount, stream);
    b.ordCap.reserve(uint64_t(taskCount) + 1, stream);
    b.scanTemp64.reserve(scanTempElements(uint64_t(taskCount) + 1), stream);
    b.results.reserve(taskCount, stream); b.ends.reserve(taskCount, stream);
    b.counters.reserve(16, stream); b.dpCells.reserve(16, stream);
    HIP_CHECK(hipMemsetAsync(b.counters.data() + 1, 0, DP_CLASSES * sizeof(uint32_t), stream));
    HIP_CHECK(hipMemsetAsync(b.dpCells.data(), 0, 16 * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(dpSizeKernel, dim3(divUp(uint64_t(taskCount) + 1, 256)), dim3(256), 0, stream,
        in.tasks, in.pairs, taskCount,
        b.dpKeysA.data(), b.dpIdsA.data(), b.ordCap.data(), b.counters.data() + 1, b.dpCells.data());
    exclusiveScan<uint64_t>(b.ordCap.data(), b.ordCap.data(), uint64_t(taskCount) + 1, b.scanTemp64.data(), stream);
    const bool inB = radixSort<uint32_t, uint32_t, true>(b.dpKeysA.data(), b.dpKeysB.data(), b.dpIdsA.data(), b.dpIdsB.data(),
        taskCount, 27, *ws.sortWs, stream);
    const uint32_t* sortedKeys = inB ? b.dpKeysB.data() : b.dpKeysA.data();
    const uint32_t* sortedIds = inB ? b.dpIdsB.data() : b.dpIdsA.data();
    f.sortedIds = sortedIds;
    HIP_CHECK(hipGetLastError());
    uint32_t* classCounts = f.classCounts;
    unsigned long long* sums = f.sums;
    HIP_CHECK(hipMemcpyAsync(classCounts, b.counters.data() + 1, sizeof(f.classCounts), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(sums, b.dpCells.data(), sizeof(f.sums), hipMemcpyDeviceToHost, stream));
    const uint64_t ordTotal = readDevice(b.ordCap.data() + taskCount, stream);      // synchronises
    DpClassLayout layout;
    layout.taskStart[0] = 0; layout.bundleStart[0] = 0;
    for(int c = 0; c < DP_CLASSES; c++) {
        const uint32_t T = 64u / uint32_t(dpLanes(c));
        layout.taskStart[c + 1] = layout.taskStart[c] + classCounts[c];
        layout.bundleStart[c + 1] = layout.bundleStart[c] + (classCounts[c] + T - 1) / T;
    }
    MI355X_ASSERT(layout.taskStart[DP_CLASSES] == taskCount);
    const uint32_t bundleTotal = layout.bundleStart[DP_CLASSES];
    b.bundleWords.reserve(uint64_t(bundleTotal) + 1, stream);
    b.scanTemp64.reserve(scanTempElements(uint64_t(bundleTotal) + 1), stream);
    hipLaunchKernelGGL(dpBundleKernel, dim3(divUp(uint64_t(bundleTotal) + 1, 256)), dim3(256), 0, stream,
        sortedKeys, layout, b.bundleWords.data());
    exclusiveScan<uint64_t>(b.bundleWords.data(), b.bundleWords.data(), uint64_t(bundleTotal) + 1, b.scanTemp64.data(), stream);
    // sums[1] bounds the trace (a bundle needs no more than the sum over its tasks): no read-back.
    b.trace.reserve(sums[1] + 64, stream);
    if(reserveOrdinals) b.ordScratch.reserve(2 * ordTotal + 2, stream);

    // Wide bands (classes 3-5: few tasks, one wavefront each) go to the side stream, widest first;
    // the narrow classes run on the main stream meanwhile.
    const bool fork = ws.wide != nullptr && ev != nullptr && (classCounts[3] || classCounts[4] || classCounts[5]);
    hipStream_t wideStream = fork ? ws.wide : stream;
    if(fork) { HIP_CHECK(hipEventRecord(ev->fork, stream)); HIP_CHECK(hipStreamWaitEvent(ws.wide, ev->fork, 0)); }
    auto timed = [&](int cls, hipStream_t st, auto launch) {
        if(ev) HIP_CHECK(hipEventRecord(ev->start[cls], st));
        launch(st);
        if(ev) HIP_CHECK(hipEventRecord(ev->stop[cls], st));
    };
    timed(5, wideStream, [&](hipStream_t st) { launchDpForward<64, 16>(in, st, b, sortedIds, layout, 5, version); });
    timed(4, wideStream, [&](hipStream_t st) { launchDpForward<64, 8>(in, st, b, sortedIds, layout, 4, version); });
    timed(3, wideStream, [&](hipStream_t st) { launchDpForward<64, 4>(in, st, b, sortedIds, layout, 3, version); });
    if(fork) HIP_CHECK(hipEventRecord(ev->join, ws.wide));
    timed(1, stream, [&](hipStream_t st) { launchDpForward<32, 2>(in, st, b, sortedIds, layout, 1, version); });
    timed(2, stream, [&](hipStream_t st) { launchDpForward<64, 2>(in, st, b, sortedIds, layout, 2, version); });
    timed(0, stream, [&](hipStream_t st) { launchDpForward<16, 2>(in, st, b, sortedIds, layout, 0, version); });
    if(fork) HIP_CHECK(hipStreamWaitEvent(stream, ev->join, 0));
    return f;
}

uint64_t runDpTasks(Context& ctx, const WorkStream& ws, BatchScratch& b, uint32_t taskCount, const DeviceOptions& opt,
    DpEvents* ev, DpBatchStats* stats)
{
    hipStream_t stream = ws.stream;
    const DpInput in{ctx.kmerIds.data(), b.pairs.data(), b.tasks.data()};
    const DpForwardState f = runDpForward(ws, b, in, taskCount, true, ev);
    if(ev) HIP_CHECK(hipEventRecord(ev->start[DP_CLASSES], stream));
    // 256-byte trace chunks: 8 iterations of the narrow classes, one iteration of the widest class.
    hipLaunchKernelGGL(dpTracebackKernel<32>, dim3(divUp(taskCount, 256)), dim3(256), 0, stream,
        in.pairs, in.tasks, f.sortedIds, taskCount,
        (const DpEnd*)b.ends.data(), (const uint64_t*)b.trace.data(),
        (const uint64_t*)b.ordCap.data(), b.ordScratch.data(), b.results.data(), opt, b.pairBest.data());
    HIP_CHECK(hipGetLastError());
    if(ev) HIP_CHECK(hipEventRecord(ev->stop[DP_CLASSES], stream));
    if(stats) for(int c = 0; c < DP_CLASSES; c++) { stats->cells[c] = f.sums[2 + c]; stats->bytes[c] = f.sums[8 + c]; stats->tasks[c] = f.classCounts[c]; }
    return f.sums[0];
}

// The two forward kernels on a batch of synthetic tasks (every band class, sequences over a small
// alphabet so that score ties are everywhere): true when every DpResult and every ordinal agrees.
thread_local int dpForwardOverride = 0;
bool dpForwardSelfTest()
{
    int device = 0;
    HIP_CHECK(hipGetDevice(&device));
    Context ctx(device);
    const int widths[DP_CLASSES] = {24, 50, 100, 200, 400, 800};
    std::vector<uint32_t> all;
    std::vector<PairDesc> pairs;
    std::vector<DpTask> tasks;
    uint64_t x = 0x9e3779b97f4a7c15ULL;
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return uint32_t(x >> 20); };
    for(int cls = 0; cls < DP_CLASSES; cls++) {
        for(int rep = 0; rep < 2; rep++) {
            // Two noisy copies of one sequence, the second one shifted: an overlap alignment near diagonal `shift`.
            const uint32_t n = uint32_t(widths[cls]) + 260u + next() % 200u, shift = next() % 150u;
            std::vector<uint32_t> base(n + shift);
            for(auto& v : base) v = next() % 7u;
            PairDesc pd; pd.begin0 = all.size();
            for(uint32_t i = 0; i < n; i++) { if(next() % 16u == 0) continue; all.push_back(next() % 11u == 0 ? next() % 7u : base[i]); }
            pd.nx = uint32_t(all.size() - pd.begin0); pd.begin1 = all.size();
            for(uint32_t i = shift; i < n + shift; i++) { if(next() % 16u == 0) continue; all.push_back(next() % 11u == 0 ? next() % 7u : base[i]); }
            pd.ny = uint32_t(all.size() - pd.begin1);
            DpTask t; t.pair = uint32_t(pairs.size()); t.label = 0;
            t.bandMin = -int32_t(shift) - widths[cls] / 2 + (rep ? 7 : 0); t.bandMax = t.bandMin + widths[cls] - 1;
            pairs.push_back(pd); tasks.push_back(t);
        }
    }
    const uint32_t taskCount = uint32_t(tasks.size());
    std::vector<uint64_t> toc = {0, all.size() / 2, all.size()};
    ctx.setMarkers(1, toc.data(), nullptr, all.data(), nullptr);
    hipStream_t stream = ctx.stream;
    DeviceOptions opt;
    std::memset(&opt, 0, sizeof(opt));
    opt.deltaX = 200; opt.deltaY = 10; opt.maxSkip = opt.maxDrift = opt.maxTrim = ~0ULL; opt.maxBand = 1024;
    const WorkStream ws{ctx.stream, &ctx.sortWs, nullptr};
    std::vector<DpResult> results[2];
    std::vector<uint32_t> ordinals[2];
    for(int version = 1; version <= 2; version++) {
        BatchScratch b;
        b.pairs.reserve(pairs.size(), stream); b.tasks.reserve(taskCount, stream); b.pairBest.reserve(pairs.size(), stream);
        HIP_CHECK(hipMemcpyAsync(b.pairs.data(), pairs.data(), pairs.size() * sizeof(PairDesc), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(b.tasks.data(), tasks.data(), taskCount * sizeof(DpTask), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemsetAsync(b.pairBest.data(), 0, 8 * pairs.size(), stream));
        dpForwardOverride = version;
        try { (void)runDpTasks(ctx, ws, b, taskCount, opt, nullptr, nullptr); } catch(...) { dpForwardOverride = 0; throw; }
        dpForwardOverride = 0;
        std::vector<DpResult>& r = results[version - 1];
        r.resize(taskCount);
        HIP_CHECK(hipMemcpyAsync(r.data(), b.results.data(), taskCount * sizeof(DpResult), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        for(const DpResult& d : r) {
            std::vector<uint32_t> o(2 * size_t(d.markerCount));
            if(d.markerCount) HIP_CHECK(hipMemcpy(o.data(), b.ordScratch.data() + 2 * d.ordBegin, 8ULL * d.markerCount, hipMemcpyDeviceToHost));
            ordinals[version - 1].insert(ordinals[version - 1].end(), o.begin(), o.end());
        }
    }
    bool aligned = false;
    for(uint32_t i = 0; i < taskCount; i++) {
        const DpResult& p = results[0][i]; const DpResult& q = results[1][i];
        if(p.markerCount != q.markerCount || p.score != q.score || p.ordBegin != q.ordBegin || p.first0 != q.first0 || p.first1 != q.first1 ||
            p.last0 != q.last0 || p.last1 != q.last1 || p.sumOffset != q.sumOffset || p.maxSkip != q.maxSkip || p.maxDrift != q.maxDrift) return false;
        aligned = aligned || p.markerCount > 100;
    }
    return aligned && ordinals[0] == ordinals[1];
}

int chooseDpForwardVersion()
{
    if(dpForwardOverride) return dpForwardOverride;
    static std::atomic<int> choice{0};
    static std::mutex mutex;
    int v = choice.load();
    if(v) return v;
    std::lock_guard<std::mutex> lock(mutex);
    v = choice.load();
    if(v) return v;
    if(const char* e = std::getenv("SHASTA_MI355X_DP_FORWARD")) {
        v = std::atoi(e) == 1 ? 1 : 2;
    } else if(dpForwardSelfTest()) {
        v = 2;
    } else {
        std::fprintf(stderr, "shasta_mi355x: the two forward DP kernels disagree on this device; using the first version "
            "(set SHASTA_MI355X_DP_FORWARD=2 to force the second).\n");
        v = 1;
    }
    choice.store(v);
    return v;
}

struct BatchOutput {
    std::vector<shasta_alignment_data> rows;
    std::vector<uint64_t> tocEnds, ordToc;
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> ordinals;
    uint64_t dpCells = 0, kmerIdBytes = 0, alignedBytes = 0, dpLaunches = 0;
    double dpSeconds = 0, forwardSeconds[DP_CLASSES] = {0}, tracebackSeconds = 0;
    DpBatchStats dpStats;
    bool hadTasks = false;
};

// Results of a "borrowed" call live here, in the context, and are reused by the next call: no
// half-gigabyte malloc / page-fault / munmap cycle per call.
struct AlignStore {
    std::vector<BatchOutput> outputs;
    std::vector<shasta_alignment_data> rows;
    std::vector<uint64_t> compressedToc, ordinalsToc;
    std::vector<uint8_t> bytes, status;
    std::vector<uint32_t> ordinals;
};

// Align method 3: what alignOrientedReads3 needs besides the outer filters.
struct Align3Plan { uint32_t k, hashThreshold; int32_t bandExtend, maxBand; };

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
    const uint64_t BATCH = 1ULL << 17;
    const uint64_t batchCount = (candidateCount + BATCH - 1) / BATCH;

    if(borrowed && !ctx.alignStore) ctx.alignStore = std::make_shared<AlignStore>();
    AlignStore localStore;
    AlignStore& store = borrowed ? *static_cast<AlignStore*>(ctx.alignStore.get()) : localStore;
    std::vector<BatchOutput>& outputs = store.outputs;
    if(outputs.size() < batchCount) outputs.resize(batchCount);
    for(uint64_t k = 0; k < batchCount; k++) {
        BatchOutput& o = outputs[k];
        o.dpCells = o.kmerIdBytes = o.alignedBytes = o.dpLaunches = 0; o.dpSeconds = o.tracebackSeconds = 0; o.hadTasks = false;
        for(int c = 0; c < DP_CLASSES; c++) o.forwardSeconds[c] = 0;
        o.dpStats = DpBatchStats();
        o.ordToc.clear(); o.ordinals.clear();
    }
    std::vector<uint8_t>& outStatus = store.status;
    outStatus.resize(std::max<uint64_t>(1, candidateCount));

    // Two host workers, each with its own stream and grow-only scratch kept in the context, take
    // the batches alternately: one worker's host-side preparation and result copies overlap the
    // other's kernels.
    struct Worker {
        hipStream_t stream = nullptr;
        RadixSortWorkspace* sortWs = nullptr;
        BatchScratch* scratch = nullptr;
        hipStream_t wide = nullptr;
        DpEvents ev;
        std::vector<PairDesc> hostPairs;
        std::vector<uint64_t> hostToc64;
        std::string error;
    };
    const int workerCount = batchCount > 1 ? 2 : 1;
    Worker workers[2];
    if(!ctx.stream2) HIP_CHECK(hipStreamCreateWithFlags(&ctx.stream2, hipStreamNonBlocking));
    for(int k = 0; k < 2; k++) {
        if(!ctx.alignScratch[k]) ctx.alignScratch[k] = std::make_shared<BatchScratch>();
        workers[k].stream = k == 0 ? ctx.stream : ctx.stream2;
        workers[k].sortWs = k == 0 ? &ctx.sortWs : &ctx.sortWs2;
        workers[k].scratch = static_cast<BatchScratch*>(ctx.alignScratch[k].get());
        if(!ctx.wideStream[k]) HIP_CHECK(hipStreamCreateWithFlags(&ctx.wideStream[k], hipStreamNonBlocking));
        workers[k].wide = ctx.wideStream[k];
        workers[k].ev.create();
    }
    hipEvent_t evBegin, evEnd, evOther;
    HIP_CHECK(hipEventCreate(&evBegin)); HIP_CHECK(hipEventCreate(&evEnd)); HIP_CHECK(hipEventCreate(&evOther));
    HIP_CHECK(hipEventRecord(evBegin, ctx.stream));

    auto processBatch = [&](Worker& w, uint64_t batchIndex) {
        hipStream_t stream = w.stream;
        const WorkStream ws{w.stream, w.sortWs, w.wide};
        BatchScratch& b = *w.scratch;
        BatchOutput& out = outputs[batchIndex];
        std::vector<PairDesc>& hostPairs = w.hostPairs;
        std::vector<uint64_t>& hostToc64 = w.hostToc64;
        const uint64_t batchBegin = batchIndex * BATCH;
        const uint32_t n = uint32_t(std::min<uint64_t>(BATCH, candidateCount - batchBegin));
        hostPairs.resize(n);
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
            MI355X_ASSERT(nx < (1ULL << 30) && ny < (1ULL << 30));
            pd.nx = uint32_t(nx); pd.ny = uint32_t(ny);
            hostPairs[k] = pd;
            out.kmerIdBytes += 4 * (nx + ny);
        }
        // Room for the DP tasks of the batch; the stage runs again with the exact count if it is short.
        // SHASTA_MI355X_INITIAL_TASKS overrides the first guess (tests use it to force the second run).
        uint32_t taskCapacity = 8 * n + 1024;
        if(const char* e = std::getenv("SHASTA_MI355X_INITIAL_TASKS")) taskCapacity = uint32_t(std::max(1L, std::atol(e)));
        b.pairs.reserve(n, stream); b.candidates.reserve(n, stream); b.tasks.reserve(taskCapacity, stream);
        b.counters.reserve(16, stream); b.pairFlags.reserve(n, stream); b.pairTie.reserve(n, stream); b.status.reserve(n, stream);
        b.pairBest.reserve(n, stream); b.dpCells.reserve(1, stream); b.pairWinner.reserve(n, stream);
        b.storedFlags.reserve(n + 1, stream); b.storedIndex.reserve(n + 1, stream);
        b.scanTemp32.reserve(scanTempElements(uint64_t(n) + 1), stream);
        b.ordCounts.reserve(n + 1, stream); b.sizes.reserve(n + 1, stream);
        b.rows.reserve(n, stream); b.rowsOut.reserve(n, stream); b.compressedToc.reserve(n + 1, stream);
        HIP_CHECK(hipMemcpyAsync(b.pairs.data(), hostPairs.data(), n * sizeof(PairDesc), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(b.candidates.data(), candidates + batchBegin, n * sizeof(shasta_oriented_read_pair), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemsetAsync(b.counters.data(), 0, 8 * sizeof(uint32_t), stream));
        HIP_CHECK(hipMemsetAsync(b.pairFlags.data(), 0, n, stream));
        HIP_CHECK(hipMemsetAsync(b.pairTie.data(), 0, n, stream));
        HIP_CHECK(hipMemsetAsync(b.pairBest.data(), 0, n * sizeof(unsigned long long), stream));
        HIP_CHECK(hipMemsetAsync(b.pairWinner.data(), 0, n * sizeof(uint32_t), stream));
        HIP_CHECK(hipMemsetAsync(b.dpCells.data(), 0, sizeof(unsigned long long), stream));

        uint32_t taskCount = 0;
        for(;;) {
        if(m3) {
            // Method 3, step 1: every diagonal of the down-sampled pair, then the band of step 2.
            std::vector<PairDesc> dsPairs(n);
            std::vector<DpTask> tasks1;
            std::vector<WideTask> wide;
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
                } else if(diagonals <= ALIGN3_WIDE_MAX_DIAGONALS) {
                    WideTask t; t.pair = k; t.chunks = uint32_t((diagonals + 63) / 64); t.traceOffset = 0;
                    wide.push_back(t);
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
                const DpInput in{ds->kmerIds.data(), b.dsPairs.data(), b.tasks1.data()};
                const DpForwardState f = runDpForward(ws, b, in, taskCount1, false, nullptr);
                out.dpCells += f.sums[0];
                hipLaunchKernelGGL(align3BandKernel<false>, dim3(divUp(taskCount1, 256)), dim3(256), 0, stream,
                    in.pairs, (const PairDesc*)b.pairs.data(), in.tasks, f.sortedIds, taskCount1,
                    (const DpEnd*)b.ends.data(), (const WideTask*)nullptr, (const WideEnd*)nullptr,
                    (const uint64_t*)b.trace.data(), (const uint32_t*)ds->ordinals.data(),
                    m3->bandExtend, m3->maxBand, b.tasks.data(), b.counters.data());
                HIP_CHECK(hipGetLastError());
            }
            // The long pairs, a few gigabytes of trace at a time.
            const uint64_t traceWordBudget = 1ULL << 29;
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
                static std::once_flag attributeOnce;
                std::call_once(attributeOnce, [] {
                    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&align3WideDpKernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, int(3 * ALIGN3_WIDE_MAX_DIAGONALS * sizeof(int32_t))));
                });
                hipLaunchKernelGGL(align3WideDpKernel, dim3(count), dim3(64), ldsBytes, stream,
                    (const uint32_t*)ds->kmerIds.data(), (const PairDesc*)b.dsPairs.data(), (const WideTask*)b.wideTasks.data(), count, rowWords,
                    b.trace.data(), b.wideEnds.data());
                HIP_CHECK(hipGetLastError());
                hipLaunchKernelGGL(align3BandKernel<true>, dim3(divUp(count, 256)), dim3(256), 0, stream,
                    (const PairDesc*)b.dsPairs.data(), (const PairDesc*)b.pairs.data(), (const DpTask*)nullptr, (const uint32_t*)nullptr, count,
                    (const DpEnd*)nullptr, (const WideTask*)b.wideTasks.data(), (const WideEnd*)b.wideEnds.data(),
                    (const uint64_t*)b.trace.data(), (const uint32_t*)ds->ordinals.data(),
                    m3->bandExtend, m3->maxBand, b.tasks.data(), b.counters.data());
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipStreamSynchronize(stream));
                begin = end;
            }
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
            // Class of a candidate: table of the tabled read at load <= 1/2, cell table sized for the
            // expected number of distinct cells (random background ~ nx*ny / alphabet, plus the
            // diagonal) at load <= 3/4.  Overflow is detected on the device and climbs one class.
            // The packed LDS cell word counts up to 2^CELLS_COUNT_BITS - 1 entries; a cell holds at most
            // ceil(deltaX * deltaY / 2) (one (x,y) per lattice point of the right parity).
            const bool packedOk = (uint64_t(opt.deltaX) * opt.deltaY + 1) / 2 < (1ULL << CELLS_COUNT_BITS) && opt.deltaX >= 2 && opt.deltaY >= 2;
            auto classFor = [&](uint64_t tabled, uint64_t nx, uint64_t ny) -> int {
                if(nx >= 65535 || ny >= 65535 || !packedOk) return CELLS_CLASSES;
                // Cell indices must fit the packed word and the single-multiply division must be exact.
                if((nx + ny) / opt.deltaX >= (1ULL << CELLS_IX_BITS) || (nx + ny) / opt.deltaY >= (1ULL << CELLS_IY_BITS)) return CELLS_CLASSES;
                if((nx + ny) * std::max<uint64_t>(opt.deltaX, opt.deltaY) >= (1ULL << 32)) return CELLS_CLASSES;
                const uint64_t cells = (nx * ny >> 13) + (nx + ny) / 32 + 32;
                for(int c = 0; c < CELLS_CLASSES; c++) {
                    if(tabled < (1ULL << CELLS_NA_LOG2[c]) && 4 * cells <= (3ULL << CELLS_SC_LOG2[c])) return c;
                }
                return CELLS_CLASSES;
            };
            std::vector<int> pairClass(n);
            std::vector<CellsChunk> classChunks[CELLS_CLASSES][2];     // [class][0 = one wave, 1 = shared table]
            std::vector<uint32_t> members;                             // candidate indices, chunk after chunk
            members.reserve(n);
            auto addChunk = [&](const uint32_t* list, uint32_t count, bool swapped, int c) {
                CellsChunk ch; ch.firstMember = uint32_t(members.size()); ch.count = uint16_t(count); ch.swapped = swapped ? 1 : 0;
                ch.naLog2 = uint32_t(CELLS_NA_LOG2[c]); ch.scLog2 = uint32_t(CELLS_SC_LOG2[c]);
                members.insert(members.end(), list, list + count);
                const int kind = (CELLS_SHARED_WAVES[c] > 1 && count >= CELLS_SHARE_MIN) ? 1 : 0;
                classChunks[c][kind].push_back(ch);
            };
            // Every candidate tables whichever of its two reads lands in the smaller class (ties: read
            // 0) and is grouped with the other candidates that table the same oriented read.
            {
                struct Keyed { uint64_t tabled; uint32_t pair; uint8_t cls, swapped; };
                std::vector<Keyed> keyed;
                keyed.reserve(n);
                for(uint32_t q = 0; q < n; q++) {
                    const PairDesc& pd = hostPairs[q];
                    const int c0 = classFor(pd.nx, pd.nx, pd.ny);
                    const int c1 = pd.ny < pd.nx ? classFor(pd.ny, pd.nx, pd.ny) : CELLS_CLASSES;
                    const bool sw = c1 < c0;
                    const int c = sw ? c1 : c0;
                    pairClass[q] = c;
                    if(c == CELLS_CLASSES) { bigList.push_back(q); bigLog2.push_back(estimateLog2(q)); continue; }
                    Keyed kd; kd.tabled = sw ? pd.begin1 : pd.begin0; kd.pair = q; kd.cls = uint8_t(c); kd.swapped = sw ? 1 : 0;
                    keyed.push_back(kd);
                }
                std::sort(keyed.begin(), keyed.end(), [](const Keyed& a, const Keyed& b) {
                    if(a.swapped != b.swapped) return a.swapped < b.swapped;
                    if(a.tabled != b.tabled) return a.tabled < b.tabled;
                    if(a.cls != b.cls) return a.cls < b.cls;
                    return a.pair < b.pair;
                });
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
                for(int c = 0; c < CELLS_CLASSES; c++) for(int kind = 0; kind < 2; kind++) {
                    uint64_t pairsIn = 0, sw = 0;
                    for(const CellsChunk& ch : classChunks[c][kind]) { pairsIn += ch.count; sw += ch.swapped; }
                    std::fprintf(stderr, "cells: class %d kind %d: %zu chunks, %llu candidates, %llu swapped\n", c, kind,
                        classChunks[c][kind].size(), (unsigned long long)pairsIn, (unsigned long long)sw);
                }
                std::fprintf(stderr, "cells: HBM-scratch list %zu\n", bigList.size());
            }
            // deltaX, deltaY >= 2 here (packedOk fails for 1 x anything >= 2046... and d = 1 gives magic 2^32): guard.
            const uint32_t magicX = uint32_t(std::min<uint64_t>((1ULL << 32) / opt.deltaX + 1, 0xffffffffULL));
            const uint32_t magicY = uint32_t(std::min<uint64_t>((1ULL << 32) / opt.deltaY + 1, 0xffffffffULL));
            // Every candidate is a member once, plus once per class it climbs to after an overflow.
            const uint64_t memberCapacity = uint64_t(CELLS_CLASSES) * n + 16;
            b.pairList.reserve(memberCapacity, stream);
            size_t membersUploaded = 0;
            for(int round = 0; round < CELLS_CLASSES; round++) {
                bool any = false;
                if(members.size() > membersUploaded) {
                    MI355X_ASSERT(members.size() <= memberCapacity);
                    HIP_CHECK(hipMemcpyAsync(b.pairList.data() + membersUploaded, members.data() + membersUploaded,
                        (members.size() - membersUploaded) * 4, hipMemcpyHostToDevice, stream));
                    membersUploaded = members.size();
                }
                for(int c = 0; c < CELLS_CLASSES; c++) {
                    std::vector<CellsChunk>& single = classChunks[c][0];
                    std::vector<CellsChunk>& shared = classChunks[c][1];
                    if(single.empty() && shared.empty()) continue;
                    any = true;
                    b.chunks.reserve(single.size() + shared.size(), stream);
                    if(!single.empty()) HIP_CHECK(hipMemcpyAsync(b.chunks.data(), single.data(), single.size() * sizeof(CellsChunk), hipMemcpyHostToDevice, stream));
                    if(!shared.empty()) HIP_CHECK(hipMemcpyAsync(b.chunks.data() + single.size(), shared.data(), shared.size() * sizeof(CellsChunk), hipMemcpyHostToDevice, stream));
                    launchCellsChunks(ctx, ws, b, c, CELLS_SHARED_WAVES[c], b.chunks.data() + single.size(), uint32_t(shared.size()), opt, magicX, magicY, taskCapacity);
                    launchCellsChunks(ctx, ws, b, c, 1, b.chunks.data(), uint32_t(single.size()), opt, magicX, magicY, taskCapacity);
                    HIP_CHECK(hipStreamSynchronize(stream));      // the lists are reused below
                    single.clear(); shared.clear();
                }
                if(!any) break;
                // Candidates that overflowed their tables climb one class.
                HIP_CHECK(hipMemcpyAsync(hostFlags.data(), b.pairFlags.data(), n, hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
                bool retry = false;
                uint64_t reasonHistogram[16] = {0};
                for(uint32_t k = 0; k < n; k++) {
                    if((hostFlags[k] & 0x0f) != PAIR_RESOURCE) continue;
                    if(debug) ++reasonHistogram[hostFlags[k] >> 4];
                    retry = true;
                    hostFlags[k] = 0;
                    // Retry alone in the next class whose table holds one of the two reads
                    // (read 0 if both fit).
                    int c = pairClass[k] + 1;
                    bool sw = false;
                    for(; c < CELLS_CLASSES; c++) {
                        const uint64_t cap = 1ULL << CELLS_NA_LOG2[c];
                        if(hostPairs[k].nx < cap) { sw = false; break; }
                        if(hostPairs[k].ny < cap) { sw = true; break; }
                    }
                    pairClass[k] = c;
                    if(c >= CELLS_CLASSES) { bigList.push_back(k); bigLog2.push_back(estimateLog2(k)); }
                    else addChunk(&k, 1, sw, c);
                }
                if(!retry) break;
                if(debug) {
                    for(int c = 0; c < CELLS_CLASSES; c++) std::fprintf(stderr, "cells: round %d retries -> class %d: %zu\n", round, c, classChunks[c][0].size());
                    std::fprintf(stderr, "cells: round %d HBM-scratch list now %zu; reasons cell-table %llu kept-list %llu geometry %llu both %llu tabled-read %llu\n", round, bigList.size(),
                        (unsigned long long)reasonHistogram[1], (unsigned long long)reasonHistogram[2], (unsigned long long)reasonHistogram[4],
                        (unsigned long long)(reasonHistogram[3] + reasonHistogram[5] + reasonHistogram[6] + reasonHistogram[7]), (unsigned long long)reasonHistogram[8]);
                }
                HIP_CHECK(hipMemcpyAsync(b.pairFlags.data(), hostFlags.data(), n, hipMemcpyHostToDevice, stream));
            }
            HIP_CHECK(hipMemcpyAsync(hostFlags.data(), b.pairFlags.data(), n, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            b.pairList.reserve(n, stream);
            const uint64_t scratchWordCap = 1ULL << 31;             // 8 GiB of HBM scratch per launch
            while(!bigList.empty()) {
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
                    hipLaunchKernelGGL(align4CellsKernel<true>, dim3(count), dim3(CELLS_THREADS), 0, stream,
                        (const uint32_t*)ctx.kmerIds.data(), (const PairDesc*)b.pairs.data(), (const uint32_t*)b.pairList.data(), count, opt,
                        b.tasks.data(), b.counters.data(), taskCapacity, b.pairFlags.data(),
                        b.bigScratch.data(), (const uint64_t*)b.bigOffsets.data(), (const uint8_t*)b.bigLog2.data());
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
        taskCount = readDevice(b.counters.data(), stream);
        if(taskCount <= taskCapacity) break;
        // More DP tasks than the list was sized for (many small components per candidate: low-complexity
        // reads, or options that keep nearly every cell).  The count is exact -- stores past the
        // capacity were dropped, the counter was not -- so the stage runs once more with room for all.
        if(m3) throw std::runtime_error("Align3: task list overflow.");
        if(std::getenv("SHASTA_MI355X_DEBUG")) std::fprintf(stderr, "cells: %u DP tasks exceed the capacity %u: running the stage again\n", taskCount, taskCapacity);
        taskCapacity = taskCount + 1024;
        b.tasks.reserve(taskCapacity, stream);
        HIP_CHECK(hipMemsetAsync(b.counters.data(), 0, 8 * sizeof(uint32_t), stream));
        HIP_CHECK(hipMemsetAsync(b.pairFlags.data(), 0, n, stream));
        }

        // K10: sort the tasks by (band class, length), bundle, forward DP, traceback.
        if(taskCount) {
            out.dpCells += runDpTasks(ctx, ws, b, taskCount, dpOpt, &w.ev, &out.dpStats);
            out.hadTasks = true;
            hipLaunchKernelGGL(winnerKernel, dim3(divUp(taskCount, 256)), dim3(256), 0, stream,
                (const DpTask*)b.tasks.data(), (const DpResult*)b.results.data(), taskCount,
                (const unsigned long long*)b.pairBest.data(), b.pairWinner.data(), b.pairTie.data());
            HIP_CHECK(hipGetLastError());
        } else {
            b.results.reserve(1, stream); b.ordScratch.reserve(2, stream);
        }

        // K11.
        const unsigned gp = divUp(uint64_t(n) + 1, 256);
        hipLaunchKernelGGL(finalizeKernel, dim3(gp), dim3(256), 0, stream,
            (const PairDesc*)b.pairs.data(), (const shasta_oriented_read_pair*)b.candidates.data(), n,
            (const DpResult*)b.results.data(), (const unsigned long long*)b.pairBest.data(),
            (const uint32_t*)b.pairWinner.data(), (const uint8_t*)b.pairTie.data(), (const uint8_t*)b.pairFlags.data(),
            opt, wantOrdinals ? 1 : 0, b.status.data(), b.rows.data(), b.storedFlags.data(), b.ordCounts.data());
        exclusiveScan<uint32_t>(b.storedFlags.data(), b.storedIndex.data(), uint64_t(n) + 1, b.scanTemp32.data(), stream);
        b.scanTemp64.reserve(scanTempElements(uint64_t(n) + 1), stream);
        exclusiveScan<uint64_t>(b.ordCounts.data(), b.ordCounts.data(), uint64_t(n) + 1, b.scanTemp64.data(), stream);
        const unsigned gw = divUp((uint64_t(n) + 1) * WAVE, 256);
        hipLaunchKernelGGL(compressSizeKernel, dim3(gw), dim3(256), 0, stream,
            (const uint32_t*)b.storedFlags.data(), (const DpResult*)b.results.data(), (const uint32_t*)b.pairWinner.data(),
            (const uint32_t*)b.ordScratch.data(), n, b.sizes.data());
        exclusiveScan<uint64_t>(b.sizes.data(), b.sizes.data(), uint64_t(n) + 1, b.scanTemp64.data(), stream);
        HIP_CHECK(hipGetLastError());
        const uint32_t storedCount = readDevice(b.storedIndex.data() + n, stream);
        const uint64_t ordTotalOut = readDevice(b.ordCounts.data() + n, stream);
        const uint64_t byteTotal = readDevice(b.sizes.data() + n, stream);
        b.bytes.reserve(byteTotal + 1, stream);
        hipLaunchKernelGGL(compressWriteKernel, dim3(gw), dim3(256), 0, stream,
            (const uint32_t*)b.storedFlags.data(), (const uint32_t*)b.storedIndex.data(), (const DpResult*)b.results.data(),
            (const uint32_t*)b.pairWinner.data(), (const uint32_t*)b.ordScratch.data(), n,
            (const uint64_t*)b.sizes.data(), b.bytes.data(), b.compressedToc.data(),
            (const shasta_alignment_data*)b.rows.data(), b.rowsOut.data());
        HIP_CHECK(hipGetLastError());

        // Copy this batch's results out (assembled in candidate order once every batch is done).
        // Device -> pinned staging (asynchronous, PCIe speed) -> the batch's output vectors.
        void* pinRows = b.pinRows.reserve(storedCount * sizeof(shasta_alignment_data));
        void* pinToc = b.pinToc.reserve(storedCount * sizeof(uint64_t));
        void* pinBytes = b.pinBytes.reserve(byteTotal);
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
        out.bytes.assign(static_cast<const uint8_t*>(pinBytes), static_cast<const uint8_t*>(pinBytes) + byteTotal);
        std::memcpy(outStatus.data() + batchBegin, pinStatus, n);
        if(wantOrdinals) {
            out.ordToc.assign(static_cast<const uint64_t*>(pinOrdToc), static_cast<const uint64_t*>(pinOrdToc) + uint64_t(n) + 1);
            out.ordinals.assign(static_cast<const uint32_t*>(pinOrdinals), static_cast<const uint32_t*>(pinOrdinals) + 2 * ordTotalOut);
        }
        HIP_CHECK(hipStreamSynchronize(stream));
        // CSR of CompressedAlignments: end offset of each stored alignment, relative to this batch.
        out.tocEnds.resize(storedCount);
        for(uint32_t k = 0; k < storedCount; k++) out.tocEnds[k] = (k + 1 < storedCount ? hostToc64[k + 1] : byteTotal);
        for(uint32_t k = 0; k < storedCount; k++) out.alignedBytes += 8ULL * out.rows[k].info.markerCount;
        if(taskCount) {
            float ms = 0;
            for(int c = 0; c < DP_CLASSES; c++) {
                if(!out.dpStats.tasks[c]) continue;
                HIP_CHECK(hipEventElapsedTime(&ms, w.ev.start[c], w.ev.stop[c]));
                out.forwardSeconds[c] = ms * 1e-3;
                out.dpSeconds += ms * 1e-3;
                ++out.dpLaunches;
            }
            HIP_CHECK(hipEventElapsedTime(&ms, w.ev.start[DP_CLASSES], w.ev.stop[DP_CLASSES]));
            out.tracebackSeconds = ms * 1e-3;
            out.dpSeconds += ms * 1e-3;
            ++out.dpLaunches;
        }

    };

    std::atomic<uint64_t> nextBatch(0);
    auto workerLoop = [&](int k) {
        try {
            HIP_CHECK(hipSetDevice(ctx.device));
            for(;;) {
                const uint64_t batchIndex = nextBatch.fetch_add(1);
                if(batchIndex >= batchCount) break;
                processBatch(workers[k], batchIndex);
            }
        } catch(const std::exception& e) {
            workers[k].error = e.what();
            nextBatch.store(batchCount);
        }
    };
    if(workerCount == 2) {
        std::thread other(workerLoop, 1);
        workerLoop(0);
        other.join();
    } else {
        workerLoop(0);
    }
    HIP_CHECK(hipEventRecord(evOther, ctx.stream2));
    HIP_CHECK(hipStreamWaitEvent(ctx.stream, evOther, 0));
    HIP_CHECK(hipEventRecord(evEnd, ctx.stream));
    HIP_CHECK(hipStreamSynchronize(ctx.stream));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, evBegin, evEnd));
    result.deviceSeconds = ms * 1e-3;
    (void)hipEventDestroy(evBegin); (void)hipEventDestroy(evEnd); (void)hipEventDestroy(evOther);
    for(int k = 0; k < 2; k++) workers[k].ev.destroy();
    for(int k = 0; k < 2; k++) if(!workers[k].error.empty()) throw std::runtime_error(workers[k].error);

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
    uint64_t rowTotal = 0, byteTotalAll = 0, ordTotalAll = 0, dpCellsTotal = 0, kmerIdBytes = 0, alignedBytes = 0, dpLaunches = 0;
    double dpSeconds = 0;
    for(uint64_t k = 0; k < batchCount; k++) {
        const BatchOutput& o = outputs[k];
        rowTotal += o.rows.size(); byteTotalAll += o.bytes.size(); ordTotalAll += o.ordinals.size() / 2;
        dpCellsTotal += o.dpCells; kmerIdBytes += o.kmerIdBytes; alignedBytes += o.alignedBytes; dpLaunches += o.dpLaunches;
        dpSeconds += o.dpSeconds;
    }
    if(borrowed) {
        store.rows.resize(std::max<uint64_t>(1, rowTotal)); store.compressedToc.resize(rowTotal + 1);
        store.bytes.resize(std::max<uint64_t>(1, byteTotalAll));
        result.alignmentData = store.rows.data(); result.compressedToc = store.compressedToc.data();
        result.compressedData = store.bytes.data(); result.status = store.status.data();
        if(wantOrdinals) {
            store.ordinalsToc.resize(candidateCount + 1); store.ordinals.resize(std::max<uint64_t>(1, 2 * ordTotalAll));
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
    uint64_t rowBase = 0, byteBase = 0, ordBase = 0;
    for(uint64_t k = 0; k < batchCount; k++) {
        const BatchOutput& o = outputs[k];
        if(!o.rows.empty()) std::memcpy(result.alignmentData + rowBase, o.rows.data(), o.rows.size() * sizeof(shasta_alignment_data));
        if(!o.bytes.empty()) std::memcpy(result.compressedData + byteBase, o.bytes.data(), o.bytes.size());
        for(size_t q = 0; q < o.tocEnds.size(); q++) result.compressedToc[rowBase + q + 1] = byteBase + o.tocEnds[q];
        if(wantOrdinals) {
            if(!o.ordinals.empty()) std::memcpy(result.ordinals + 2 * ordBase, o.ordinals.data(), o.ordinals.size() * sizeof(uint32_t));
            for(size_t q = 1; q < o.ordToc.size(); q++) result.ordinalsToc[k * BATCH + q] = ordBase + o.ordToc[q];
            ordBase += o.ordinals.size() / 2;
        }
        rowBase += o.rows.size(); byteBase += o.bytes.size();
    }

    for(int c = 0; c < DP_CLASSES; c++) { ctx.times.dpForwardSeconds[c] = 0; ctx.times.dpForwardLaunches[c] = 0; ctx.times.dpForwardCells[c] = 0; ctx.times.dpForwardBytes[c] = 0; }
    ctx.times.dpTracebackSeconds = 0; ctx.times.dpTracebackLaunches = 0;
    for(uint64_t k = 0; k < batchCount; k++) {
        const BatchOutput& o = outputs[k];
        if(!o.hadTasks) continue;
        for(int c = 0; c < DP_CLASSES; c++) {
            if(!o.dpStats.tasks[c]) continue;
            ctx.times.dpForwardSeconds[c] += o.forwardSeconds[c]; ctx.times.dpForwardLaunches[c] += 1;
            ctx.times.dpForwardCells[c] += o.dpStats.cells[c]; ctx.times.dpForwardBytes[c] += o.dpStats.bytes[c];
        }
        ctx.times.dpTracebackSeconds += o.tracebackSeconds; ctx.times.dpTracebackLaunches += 1;
    }
    ctx.times.alignDpSeconds = dpSeconds;
    ctx.times.alignDpLaunches = dpLaunches;
    ctx.times.alignDpCells = dpCellsTotal;
    ctx.times.alignBytes = kmerIdBytes + alignedBytes;
    result.alignmentCount = rowTotal;
    result.dpCellCount = dpCellsTotal;
    result.kmerIdBytes = kmerIdBytes;
    result.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

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
    // The DP kernels carry the scores as compile-time constants (the values every shipped
    // configuration uses for method 3 and the ones method 4 hard-wires).
    if(o.matchScore != MATCH_SCORE || o.mismatchScore != MISMATCH_SCORE || o.gapScore != GAP_SCORE) {
        throw std::runtime_error("Align3: only matchScore 6, mismatchScore -1, gapScore -1 are supported.");
    }
    if(o.k < 1 || o.k > 16) throw std::runtime_error("Align3: k must be in [1, 16].");
    if(!(o.downsamplingFactor >= 0. && o.downsamplingFactor <= 1.)) throw std::runtime_error("Align3: downsamplingFactor must be in [0, 1].");
    if(o.bandExtend < 0 || o.bandExtend > (1 << 20)) throw std::runtime_error("Align3: bandExtend must be in [0, 2^20].");
    if(o.maxBand < 0 || o.maxBand > 1023) throw std::runtime_error("Align3: maxBand must be in [0, 1023] (1024 diagonals per DP).");
    Align3Plan plan;
    plan.k = uint32_t(o.k);
    plan.hashThreshold = uint32_t(o.downsamplingFactor * double(std::numeric_limits<uint32_t>::max()));   // src/AssemblerAlign3.cpp:71-72
    plan.bandExtend = int32_t(o.bandExtend); plan.maxBand = int32_t(o.maxBand);
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

// Unit seam: one banded DP on the device (used by the parity tests of K10 alone).
int dpForwardVersion() { return chooseDpForwardVersion(); }

// Unit seam: K10 on many (pair, band) tasks at once -- sorted, bundled and run exactly as the tasks of an Align4 batch
// are, so that wavefronts hold several tasks of different geometry.  Task t aligns kmerIds[begin0[t] .. +nx[t]) with
// kmerIds[begin1[t] .. +ny[t]) inside [bandMin[t], bandMax[t]].  Results in task order: counts[t] aligned pairs,
// scores[t], and the pairs themselves concatenated in ordinals.
void bandedDpManyUnit(const uint32_t* kmerIds, uint64_t kmerCount, uint64_t taskCount,
    const uint64_t* begin0, const uint32_t* nx, const uint64_t* begin1, const uint32_t* ny, const int32_t* bandMin, const int32_t* bandMax,
    uint64_t* counts, int32_t* scores, uint32_t* ordinals, uint64_t capacity)
{
    int device = 0;
    HIP_CHECK(hipGetDevice(&device));
    Context ctx(device);
    if(taskCount == 0) return;
    if(taskCount > (1u << 24)) throw std::runtime_error("banded_dp_many: too many tasks.");
    std::vector<PairDesc> pairs(taskCount);
    std::vector<DpTask> tasks(taskCount);
    for(uint64_t t = 0; t < taskCount; t++) {
        if(nx[t] == 0 || ny[t] == 0 || begin0[t] + nx[t] > kmerCount || begin1[t] + ny[t] > kmerCount) throw std::runtime_error("banded_dp_many: a sequence is empty or outside kmerIds.");
        if(bandMin[t] > bandMax[t] || bandMax[t] - bandMin[t] + 1 > 1024) throw std::runtime_error("banded_dp_many: band width must be in [1, 1024].");
        if(bandMin[t] > int32_t(nx[t]) || bandMax[t] < -int32_t(ny[t])) throw std::runtime_error("banded_dp_many: the band misses the matrix.");
        pairs[t].begin0 = begin0[t]; pairs[t].begin1 = begin1[t]; pairs[t].nx = nx[t]; pairs[t].ny = ny[t];
        tasks[t].pair = uint32_t(t); tasks[t].bandMin = bandMin[t]; tasks[t].bandMax = bandMax[t]; tasks[t].label = 0;
    }
    std::vector<uint64_t> toc = {0, kmerCount / 2, kmerCount};
    ctx.setMarkers(1, toc.data(), nullptr, kmerIds, nullptr);
    hipStream_t stream = ctx.stream;
    BatchScratch b;
    b.pairs.reserve(taskCount, stream); b.tasks.reserve(taskCount, stream); b.pairBest.reserve(taskCount, stream);
    HIP_CHECK(hipMemcpyAsync(b.pairs.data(), pairs.data(), taskCount * sizeof(PairDesc), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(b.tasks.data(), tasks.data(), taskCount * sizeof(DpTask), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemsetAsync(b.pairBest.data(), 0, 8 * taskCount, stream));
    DeviceOptions opt;
    std::memset(&opt, 0, sizeof(opt));
    opt.deltaX = 200; opt.deltaY = 10; opt.maxSkip = opt.maxDrift = opt.maxTrim = ~0ULL; opt.maxBand = 1024;
    const WorkStream ws{ctx.stream, &ctx.sortWs, nullptr};
    (void)runDpTasks(ctx, ws, b, uint32_t(taskCount), opt, nullptr, nullptr);
    std::vector<DpResult> results(taskCount);
    HIP_CHECK(hipMemcpyAsync(results.data(), b.results.data(), taskCount * sizeof(DpResult), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    uint64_t used = 0;
    for(uint64_t t = 0; t < taskCount; t++) {
        const DpResult& r = results[t];
        if(used + r.markerCount > capacity) throw std::runtime_error("banded_dp_many: output capacity too small.");
        if(r.markerCount) HIP_CHECK(hipMemcpy(ordinals + 2 * used, b.ordScratch.data() + 2 * r.ordBegin, 8ULL * r.markerCount, hipMemcpyDeviceToHost));
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
    b.pairs.reserve(1, stream); b.tasks.reserve(1, stream); b.pairBest.reserve(1, stream);
    HIP_CHECK(hipMemcpyAsync(b.pairs.data(), &pd, sizeof(pd), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(b.tasks.data(), &task, sizeof(task), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemsetAsync(b.pairBest.data(), 0, 8, stream));
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
