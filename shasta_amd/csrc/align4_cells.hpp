// Align4 on MI355X, K8/K9: from a candidate's two marker sequences to its DP tasks -- the sparse marker-match
// matrix in rotated (X, Y) cells, forward / backward reachability, components, one band per component
// (/root/reference/src/Align4.cpp:195-267, 380-436, 682-872).  Included by align4.hip inside its anonymous namespace
// (constants, PairDesc / DpTask / DeviceOptions are defined there).
//   align4CellsChunkKernel<Q>   candidates that share one oriented read: one LDS table of that read per workgroup
//   align4CellsKernel<BIG>      one workgroup per candidate, tables in HBM scratch: reads beyond the LDS classes
#pragma once

// ---------------------------------------------------------------------------
// K8/K9: cells.
// ---------------------------------------------------------------------------
// A few hundred long candidates per batch, a workgroup each: as many threads and as much of a CU's LDS as a workgroup can have
// (16 wavefronts, 128 KB), so that a pair of 10 k-marker reads is two passes of ten rounds instead of five passes of forty.
constexpr int CELLS_THREADS = 1024;
constexpr int MATCH_CHUNK = 8192;          // markers of read 1 hashed per round
constexpr int MATCH_SLOTS_LOG2 = 14;
constexpr int MATCH_SLOTS = 1 << MATCH_SLOTS_LOG2;
constexpr int CELL_SLOTS = 2048;
constexpr int MAX_CELLS = 1024;
constexpr uint32_t EMPTY32 = 0xffffffffu;
constexpr uint64_t EMPTY64 = ~0ULL;

constexpr uint32_t F_NEAR_LT = 1, F_NEAR_RB = 2, F_FWD = 4, F_BWD = 8;
constexpr uint8_t PAIR_RESOURCE = 1;       // a cell table overflowed: retried with a larger table in HBM
constexpr uint8_t PAIR_TOO_LONG = 2;       // outside the supported geometry (iX/iY >= 2^16): skipped + reported
constexpr int CELLS_WIDE_COUNTER = 13;     // taskCount[13]: components of more than 1024 diagonals, listed from the back of the task list

__device__ __forceinline__ uint32_t hash32(uint32_t k) { return k * 2654435761u; }

// A word of LDS that other lanes change by atomics, read NOW (the compiler must not reuse an earlier value): a relaxed atomic
// load, which stays a ds_read_b32.  A `volatile` access does not -- the address-space inference leaves volatile accesses
// alone, so `*(volatile uint32_t*)&cells[k]` was a flat_load_dword sc0 sc1 followed by s_waitcnt vmcnt(0) lgkmcnt(0): a trip
// through the flat path that also waited for every global load in flight (the next round's markers), once per counted slot,
// in rounds 2 and 3 alike (found in the ISA, scripts/isa_loop.py does not tell the two apart).
__device__ __forceinline__ uint32_t ldsLoadNow(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }

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
// DUMP (the second look at a candidate whose best components tie, see align4CellsChunkKernel): instead of DP tasks, the
// candidate's active cells at activeKeys[activeOffsets[blockIdx.x] ...] (room for half its table slots) and their number in
// activeCounts[blockIdx.x] (~0 if the tables overflowed).
template<bool BIG, bool DUMP = false>
__global__ void __launch_bounds__(CELLS_THREADS)
align4CellsKernel(
    const uint32_t* __restrict__ kmerIds, const PairDesc* __restrict__ pairs,
    const uint32_t* __restrict__ pairList, uint32_t listCount,
    DeviceOptions opt, DpTask* __restrict__ tasks, uint32_t* __restrict__ taskCount, uint32_t taskCapacity,
    uint8_t* __restrict__ pairFlags,
    uint32_t* __restrict__ bigScratch, const uint64_t* __restrict__ bigOffsets, const uint8_t* __restrict__ bigSlotsLog2,
    uint32_t* __restrict__ activeKeys, const uint64_t* __restrict__ activeOffsets, uint32_t* __restrict__ activeCounts)
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
            uint32_t slot = hash32(k) >> (32 - MATCH_SLOTS_LOG2);
            for(;;) {
                const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&matchTab[slot]), EMPTY64, entry);
                if(old == EMPTY64) break;
                slot = (slot + 1) & (MATCH_SLOTS - 1);
            }
        }
        __syncthreads();
        for(uint32_t x = tid; x < nx; x += CELLS_THREADS) {
            // (The candidate runs again in a larger table, or is reported, whatever else is counted; and every further match of a
            // FULL table walks all of it -- `slots` compare-and-swaps in device memory each -- before saying so again.)
            if(ldsLoadNow(&sOverflow) != 0u) break;
            const uint32_t k = p0[x];
            uint32_t slot = hash32(k) >> (32 - MATCH_SLOTS_LOG2);
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
    if(DUMP && (sOverflow || sCells == 0)) { if(tid == 0) activeCounts[blockIdx.x] = sOverflow ? 0xffffffffu : 0u; return; }
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
    if(DUMP) {
        __syncthreads();
        if(tid == 0) sChanged = 0;
        __syncthreads();
        uint32_t* const out = activeKeys + activeOffsets[blockIdx.x];
        for(int c = tid; c < n; c += CELLS_THREADS) {
            if(ld<BIG>(&cLabel[c]) != EMPTY32) out[atomicAdd(&sChanged, 1u)] = ld<BIG>(&cKey[c]);      // (n <= half the slots: the room the host gave)
        }
        __syncthreads();
        if(tid == 0) activeCounts[blockIdx.x] = sChanged;
        return;
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
        if(bandWidth > 1024) {
            // More diagonals than the banded DP kernels hold (Align.maxBand beyond 1024): to the BACK of the task list, for the
            // wide DP (taskCount[CELLS_WIDE_COUNTER] counts them; the host checks that the two ends did not meet).
            const uint32_t w = atomicAdd(taskCount + CELLS_WIDE_COUNTER, 1u);
            if(w < taskCapacity) { DpTask task; task.pair = pair; task.bandMin = bandMin; task.bandMax = bandMax; task.label = key; tasks[taskCapacity - 1u - w] = task; }
            continue;
        }
        const uint32_t t = atomicAdd(taskCount, 1u);
        if(t < taskCapacity) { DpTask task; task.pair = pair; task.bandMin = bandMin; task.bandMax = bandMax; task.label = key; tasks[t] = task; }
    }
}

// ---------------------------------------------------------------------------
// K8/K9, fast path.  A CHUNK is a set of candidates that share one oriented read: read 0
// (candidates arrive sorted by readId0, src/LowHash0.cpp:204-214, and read 0 is always on
// strand 0, src/AssemblerAlign.cpp:382) or, when read 1 is the shorter one, read 1 ("swapped",
// gathered by the host).  One workgroup of W wavefronts per chunk.  The whole workgroup works on ONE candidate at a time:
//   build   all wavefronts index the shared read's markers in an EXACT bucketed LDS table, once per chunk: a counting sort
//           of (rest of hash | ordinal) words by the top bits of kmerId * odd constant (a bijection: bucket + rest = the kmer
//           id, so a match needs no second look);
//   probe   a candidate's other read is streamed through the table, four markers per lane per round, every W-th round
//           by the same wavefront: a marker walks the entries of its bucket, matches wait in a queue of the wavefront;
//   count   (x,y) -> cell by magic-number division (getXY + createCells, src/Align4.cpp:171-177,380-436), one LDS
//           atomic per hit into ONE cell region shared by the wavefronts (LDS atomics work across them); the
//           increment that reaches minEntryCountPerCell appends the cell to the candidate's kept list (:417);
//   graph   after W candidates have been streamed (kept lists in W slots), wavefront w takes the w-th of them: the kept
//           cells (Q per lane) live in registers; their forward/backward adjacency is a bit mask per cell, so
//           forwardSearch / backwardSearch (:682-788) and the connected components (:792-868) are iterated ballots with
//           no memory traffic -- W graphs at a time (round 2's first cooperative kernel left the graph to wavefront 0 while
//           the others waited: with the stream W times faster, the graph had become half of a candidate's time);
//   tasks   one DP task per component (:890-934), staged in the wavefront's slot, appended with one global atomic.
// Round 1 gave each wavefront its own cell region and its own candidates (4-6 wavefronts of LDS per workgroup, 6-8
// wavefronts per CU): the kernel is bound by the latency of its dependent LDS chains, so wavefronts per CU decide.
// Candidates that overflow a table or the kept list are flagged PAIR_RESOURCE and retried in a
// larger class, finally by align4CellsKernel<true>.
// Dynamic LDS (32-bit words): range[NA << (F - 1)] (16-bit bucket starts, 2^F NA buckets, F = cellsBucketFactorLog2(Q)) | entries[NA] | cells[SC] (one byte
//   per cell of the grid, or iY | iX | count packed) | per wavefront a slot (cellsSlotLdsWords).
// ---------------------------------------------------------------------------
// firstMember indexes the member list (candidate indices of the batch).
struct CellsChunk { uint32_t firstMember; uint16_t count, swapped; uint32_t naLog2, scLog2; };      // swapped: bit 0 = read 1 is the tabled one, bit 1 = count in the packed table even if the byte grid would fit

// The matches of every candidate, listed by the cells kernel for the sparse form of the banded alignment (align4_sparse.hpp):
// hits[base[k] .. base[k + 1]) = room for candidate k's matches, (x << 16) | y in the order the wavefronts found them;
// meta[k] = how many there were (beyond the room: counted, not stored) | bit 31: the chunk tabled read 1 (the stream was read 0);
// 0xffffffff = no list (the candidate's cells were computed by the kernel with its tables in HBM scratch).  hits == nullptr: no lists.
struct HitLists { uint32_t* hits; const uint64_t* base; uint32_t* meta; };
constexpr uint32_t HIT_LIST_NONE = 0xffffffffu;
// Room for a candidate's matches: the shorter read's markers and a quarter (every one matched, some twice), the random background
// (nx ny >> matchShift, Context::matchShift: measured on a sample of the read set's markers, 13 at least -- 2^13 is about the smallest
// marker alphabet of Shasta's configurations: k = 10 at markerDensity 0.1 has about 7 900 marker k-mers, the bench's synthetic reads
// 15 000; 2^12 asked for 24 GB per batch of ultra-long pairs; k = 14 has 640 000 and gets 18) and some slack.  Repeat-rich pairs exceed it and take the dense DP (counted:
// the kernel table's "dense DP because: the candidate's match list overflowed").
__host__ __device__ inline uint32_t hitListCapacity(uint32_t nx, uint32_t ny, int matchShift)
{
    const uint32_t shorter = nx < ny ? nx : ny;
    return shorter + shorter / 4 + uint32_t((uint64_t(nx) * uint64_t(ny)) >> matchShift) + 192u;
}

#ifdef SHASTA_PROFILE_PHASES
__device__ unsigned long long g_phaseCycles[16];
#define PHASE_MARK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
    if((threadIdx.x & 63) == 0) atomicAdd(&g_phaseCycles[k], now_ - phaseT_); phaseT_ = now_; } while(0)
#define PHASE_BEGIN() unsigned long long phaseT_ = __builtin_readcyclecounter()
// Inside the stream loop: cycles of probe [7], first resolve pass [8], counting [9], further matches [10]; rounds [11],
// iterations of the further-matches loop [12], count calls [13].  (Exact table: [9] = matches queued, [11] rounds, [12] trips of
// the loop over the buckets' entries, [13] drains of a full queue, [14] drains at the end of a candidate.)
#define SUBPHASE_DECLARE() unsigned long long sub_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, subT_ = 0
#define SUBPHASE_START() subT_ = __builtin_readcyclecounter()
#define SUBPHASE_ADD(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); sub_[k] += now_ - subT_; subT_ = now_; } while(0)
#define SUBPHASE_COUNT(k) (++sub_[k])
#define SUBPHASE_COUNT_N(k, n) (sub_[k] += (n))
#define SUBPHASE_FLUSH() do { if((threadIdx.x & 63) == 0) for(int k_ = 0; k_ < 8; k_++) atomicAdd(&g_phaseCycles[7 + k_], sub_[k_]); for(int k_ = 0; k_ < 8; k_++) sub_[k_] = 0; } while(0)
#else
#define SUBPHASE_DECLARE() do {} while(0)
#define SUBPHASE_START() do {} while(0)
#define SUBPHASE_ADD(k) do {} while(0)
#define SUBPHASE_COUNT(k) do {} while(0)
#define SUBPHASE_COUNT_N(k, n) do {} while(0)
#define SUBPHASE_FLUSH() do {} while(0)
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


// Buckets of the exact table per marker the tabled read may have (log2): 2 NA buckets, load factor below 1/2 (the loop over
// a bucket's entries runs as long as the fullest bucket among a round's 256 markers: 3.7 trips per round at 100 k reads).
#ifndef SHASTA_CELLS_BUCKET_FACTOR_LOG2
#define SHASTA_CELLS_BUCKET_FACTOR_LOG2 1
#endif
static_assert(SHASTA_CELLS_BUCKET_FACTOR_LOG2 >= 1 && SHASTA_CELLS_BUCKET_FACTOR_LOG2 <= 3, "bucket starts are 16-bit halves, two to a word");
// (The class with the largest cell region, Q = 4, has no LDS to spare: two buckets per marker there whatever the others get.)
__host__ __device__ constexpr int cellsBucketFactorLog2(int Q) { return Q >= 4 ? 1 : SHASTA_CELLS_BUCKET_FACTOR_LOG2; }
#ifndef SHASTA_CELLS_DRAIN
#define SHASTA_CELLS_DRAIN 2
#endif
constexpr int CELLS_DRAIN = SHASTA_CELLS_DRAIN;      // queue entries a lane counts per pass
constexpr int CELLS_RANGE_PAD = 4;        // words behind the bucket starts: the first holds the end of the last bucket (a marker reads starts[b] and starts[b + 1])
constexpr int CELLS_UNROLL = 4;           // markers per lane per round
constexpr int CELLS_IX_BITS = 10, CELLS_IY_BITS = 12, CELLS_COUNT_BITS = 10;   // packed LDS cell word
// The windowed class (align4CellsLongKernel: either read of any length below 65 535 markers): fourteen bits of iY (nx + ny up to
// 163 000 at deltaY = 10) and EIGHT of count -- a lane adds only while the count it sees is below the threshold (the count beyond
// it means nothing: createCells only asks whether a cell has minEntryCountPerCell entries), so the field holds threshold + the adds
// in flight; an add that finds 255 has carried into the cell's key: the candidate is flagged and climbs to the HBM-scratch kernel.
constexpr int CELLS_LONG_IY_BITS = 14, CELLS_LONG_COUNT_BITS = 8;
constexpr int CELLS_STAGE = 8;            // DP tasks staged per wave before one global append
// A wavefront's slot: kept[64 Q] | queue[max(64 Q, CELLS_QUEUE)] (both the graph's cell map[128 Q] later) | scratch[8] | stage[4 CELLS_STAGE].
// The queue takes the matches of one trip of all CELLS_UNROLL groups of a round at once: at most CELLS_UNROLL * 64.
constexpr int CELLS_QUEUE = CELLS_UNROLL * 64;
__host__ __device__ inline size_t cellsQueueWords(int Q) { return size_t(64 * Q > CELLS_QUEUE ? 64 * Q : CELLS_QUEUE); }
__host__ __device__ inline size_t cellsSlotLdsWords(int Q)
{
    return 64 * size_t(Q) + cellsQueueWords(Q) + 8 + 4 * CELLS_STAGE;
}
__host__ __device__ inline size_t cellsChunkLdsWords(int naLog2, int scLog2, int Q, int waves)
{
    return ((size_t(1) << naLog2) << (cellsBucketFactorLog2(Q) - 1)) + CELLS_RANGE_PAD + (size_t(1) << naLog2) + (size_t(1) << scLog2) + size_t(waves) * cellsSlotLdsWords(Q);
}

#ifndef SHASTA_ABLATE
#define SHASTA_ABLATE 0          // timing experiments (wrong results): 1 = without the kept-cell graphs, 2 = without the stream,
                                 // 3 = without the two barriers per candidate, 4 = matches queued but not counted, 5 = without the loop over the buckets' entries
#endif
#ifndef SHASTA_CELLS_MAX_THREADS
#define SHASTA_CELLS_MAX_THREADS 384
#endif
// Wavefronts per SIMD the register allocator must make room for.  Five: 96 vector registers (the kernel wants 111 and spills
// four dwords), and the first class's 29 KB of LDS per workgroup lets five workgroups share a CU: 55 -> 50 ms solo
// (scripts/gpu_r02_call30.sh).  At the start of the round the same constraint cost 20 spilled dwords and made the kernel slower.
#ifndef SHASTA_CELLS_WAVES_PER_SIMD
#define SHASTA_CELLS_WAVES_PER_SIMD 5
#endif
#ifdef __HIPCC__
#define SHASTA_CELLS_OCCUPANCY __attribute__((amdgpu_waves_per_eu(SHASTA_CELLS_WAVES_PER_SIMD, SHASTA_CELLS_WAVES_PER_SIMD)))
#else
#define SHASTA_CELLS_OCCUPANCY            // (the wave64 emulator of tests/emu compiles this file as plain C++)
#endif
// DUMP (chunks of ONE candidate; the second look at a candidate whose best components tie on markerCount): instead of DP
// tasks the kernel leaves the candidate's active cells, activeKeys[64 Q blockIdx.x ...] and activeCounts[blockIdx.x] (~0 if the
// tables overflowed) -- the reference breaks such ties by the order of its union-find representatives, which the host
// reproduces from the set of active cells (resolveComponentTies, align4.hip).
// LONG (align4CellsLongKernel): the windowed class.  A chunk is any CELLS_LONG_WAVES candidates; each tables its SHORTER read in
// windows of 2^13 markers, one after the other -- the whole workgroup builds the window's table and streams the other read through
// it, the cell counts (the packed table with the wider fields above) staying from window to window -- so that a pair of two reads
// beyond every LDS table class (/root/reference/conf/Nanopore-UL-May2022.conf: no read below 50 000 bases) has its cells counted
// and its matches LISTED in LDS like any other, instead of in the kernel with its tables in HBM scratch, which lists nothing.
constexpr int CELLS_LONG_WAVES = 16, CELLS_LONG_THREADS = 64 * CELLS_LONG_WAVES;

// ---- the kept-cell graph of a candidate with more kept cells than a wavefront holds in registers (64 Q) ----------------------------
// A pair of two long reads with a long overlap keeps about a cell and a half per 100 markers of it in each read: beyond 256 from overlaps of
// some 20 000 markers on (1 000 candidates per step of the ultra-long shape; until round 6 they went on to the HBM-scratch kernel, which
// lists no matches, and from there to the dense DP: 570 ms of kernels per step for 0.15 % of the candidates).  One wavefront, the kept cells
// in LDS, up to CELLS_BIG_KEPT of them:
//   order     the cells by (iX, iY): a counting sort by column (iX < 2^CELLS_IX_BITS), then every cell's rank among its column's;
//   sweeps    forwardSearch / backwardSearch (src/Align4.cpp:682-788) move by at most one column, and inside a column along runs of
//             consecutive iY: ONE pass over the columns in ascending (descending) order settles them -- a column's cells in the lanes, the
//             reached cells of the column before in registers, the runs closed by shifts of a ballot.  Components (:792-868, 8-neighbourhood
//             of the active cells): the smallest position of a component's cells, passed along in alternating sweeps until one changes nothing
//             (two for a chain of cells; more only where branches meet);
//   tasks     one per component from the iY range of its cells (:890-934).
// A column of more than 64 kept cells (6 400 matches inside 200 anti-diagonals: a tandem repeat) is left to the HBM-scratch kernel.
constexpr int CELLS_BIG_KEPT = 4096;
constexpr int CELLS_BIG_COLUMNS = 1 << CELLS_IX_BITS;
// Words of LDS the graph works in (colStart | cursor | the cells in column order | state bytes | labels (16-bit) | iY ranges of the components).
constexpr int CELLS_BIG_WORK_WORDS = (CELLS_BIG_COLUMNS + 1) + CELLS_BIG_COLUMNS + CELLS_BIG_KEPT + CELLS_BIG_KEPT / 4 + CELLS_BIG_KEPT / 2 + 2 * CELLS_BIG_KEPT;
constexpr uint32_t BIG_NEAR_LT = 1, BIG_NEAR_RB = 2, BIG_FWD = 4, BIG_BWD = 8;

// keys[0 .. n): the kept cells (iY << 16 | iX), any order; afterwards in (iX, iY) order.  Returns false if the graph does not fit (the
// caller flags the candidate).  One wavefront; `work`: CELLS_BIG_WORK_WORDS words of LDS no other wavefront touches meanwhile.
__device__ inline bool cellsBigGraph(uint32_t* __restrict__ keys, int n, uint32_t* __restrict__ work, uint32_t pair, uint32_t nx, uint32_t ny, const DeviceOptions& opt,
    DpTask* __restrict__ tasks, uint32_t* __restrict__ taskCount, uint32_t taskCapacity)
{
    const int lane = laneId();
    uint32_t* const colStart = work;                                              // [COLUMNS + 1]
    uint32_t* const cursor = colStart + CELLS_BIG_COLUMNS + 1;                    // [COLUMNS]
    uint32_t* const sorted = cursor + CELLS_BIG_COLUMNS;                          // [KEPT]
    uint8_t* const state = reinterpret_cast<uint8_t*>(sorted + CELLS_BIG_KEPT);   // [KEPT]
    uint16_t* const label = reinterpret_cast<uint16_t*>(sorted + CELLS_BIG_KEPT + CELLS_BIG_KEPT / 4);          // [KEPT]
    uint32_t* const yMin = sorted + CELLS_BIG_KEPT + CELLS_BIG_KEPT / 4 + CELLS_BIG_KEPT / 2;                    // [KEPT]
    uint32_t* const yMax = yMin + CELLS_BIG_KEPT;                                 // [KEPT]
    // ---- the cells in (iX, iY) order ----
    for(int k = lane; k <= CELLS_BIG_COLUMNS; k += WAVE) colStart[k] = 0;
    waveLdsSync();
    for(int i = lane; i < n; i += WAVE) atomicAdd(&colStart[(keys[i] & 0xffffu) + 1u], 1u);
    waveLdsSync();
    bool crowded = false;
    uint32_t firstColumn = CELLS_BIG_COLUMNS, lastColumn = 0;
    {
        constexpr int PER = (CELLS_BIG_COLUMNS + 1 + WAVE - 1) / WAVE;
        const int first = lane * PER, end = min(first + PER, CELLS_BIG_COLUMNS + 1);
        uint32_t sum = 0;
        for(int k = first; k < end; k++) {
            const uint32_t m = colStart[k];                                       // cells of column k - 1
            sum += m; crowded |= m > uint32_t(WAVE);
            if(m) { firstColumn = min(firstColumn, uint32_t(k - 1)); lastColumn = max(lastColumn, uint32_t(k - 1)); }
        }
        uint32_t inclusive = sum;
#pragma unroll
        for(int d = 1; d < WAVE; d <<= 1) { const uint32_t o = uint32_t(__shfl_up(int(inclusive), d, WAVE)); if(lane >= d) inclusive += o; }
        uint32_t running = inclusive - sum;
        for(int k = first; k < end; k++) { running += colStart[k]; colStart[k] = running; }       // colStart[k] = cells of the columns below k
#pragma unroll
        for(int d = 32; d >= 1; d >>= 1) {
            firstColumn = min(firstColumn, uint32_t(__shfl_xor(int(firstColumn), d, WAVE))); lastColumn = max(lastColumn, uint32_t(__shfl_xor(int(lastColumn), d, WAVE)));
        }
    }
    if(__any(crowded)) return false;
    waveLdsSync();
    for(int k = lane; k < CELLS_BIG_COLUMNS; k += WAVE) cursor[k] = colStart[k];
    waveLdsSync();
    for(int i = lane; i < n; i += WAVE) { const uint32_t key = keys[i]; sorted[atomicAdd(&cursor[key & 0xffffu], 1u)] = key; }
    waveLdsSync();
    for(int i = lane; i < n; i += WAVE) {
        const uint32_t key = sorted[i], column = key & 0xffffu;
        const uint32_t s = colStart[column], e = colStart[column + 1u];
        uint32_t rank = 0;
        for(uint32_t j = s; j < e; j++) rank += (sorted[j] >> 16) < (key >> 16) ? 1u : 0u;
        keys[s + rank] = key;
    }
    waveLdsSync();
    // ---- boundary flags (:424-429 with the corner rules of :530-626) ----
    for(int i = lane; i < n; i += WAVE) {
        const uint32_t key = keys[i];
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
        if(uint64_t(left) < opt.maxDistanceFromBoundary || uint64_t(top) < opt.maxDistanceFromBoundary) f |= BIG_NEAR_LT;
        if(uint64_t(right) < opt.maxDistanceFromBoundary || uint64_t(bottom) < opt.maxDistanceFromBoundary) f |= BIG_NEAR_RB;
        state[i] = uint8_t(f);
        yMin[i] = EMPTY32; yMax[i] = 0;
    }
    waveLdsSync();
    // A column's cells in the lanes (m <= 64 of them, iY ascending); A: lane i and lane i + 1 hold consecutive iY.
    // closeRuns: the lanes of `reached` and every lane joined to one of them through consecutive iY.
    auto closeRuns = [](uint64_t reached, uint64_t A) {
        for(;;) {
            const uint64_t grown = reached | ((reached & A) << 1) | ((reached >> 1) & A);
            if(grown == reached) return reached;
            reached = grown;
        }
    };
    // ---- forwardSearch (:682-729), then backwardSearch (:736-787): seeds near right / bottom AND forward accessible ----
    for(int pass = 0; pass < 2; pass++) {
        const bool forward = pass == 0;
        uint64_t neighbourReached = 0;                      // reached cells of the column before (forward) / behind (backward), by lane
        uint32_t neighbourY = 0;                            // ... their iY
        for(int32_t step = 0; step <= int32_t(lastColumn) - int32_t(firstColumn); step++) {
            const uint32_t column = forward ? firstColumn + uint32_t(step) : lastColumn - uint32_t(step);
            const uint32_t s = colStart[column], m = colStart[column + 1u] - s;
            if(m == 0) { neighbourReached = 0; continue; }
            const bool valid = uint32_t(lane) < m;
            const uint32_t y = keys[s + (valid ? uint32_t(lane) : 0u)] >> 16;
            const uint32_t st = state[s + (valid ? uint32_t(lane) : 0u)];
            bool hit = false;
            for(uint64_t todo = neighbourReached; todo; todo &= todo - 1) {
                const uint32_t other = __builtin_amdgcn_readlane(neighbourY, __ffsll((unsigned long long)todo) - 1);
                hit = hit || (y + 1u - other) <= 2u;                            // |y - other| <= 1
            }
            const bool seed = forward ? (st & BIG_NEAR_LT) != 0 : ((st & BIG_NEAR_RB) != 0 && (st & BIG_FWD) != 0);
            const uint32_t above = uint32_t(__shfl_down(int(y), 1, WAVE));
            const uint64_t A = __ballot(valid && uint32_t(lane) + 1u < m && above == y + 1u);
            const uint64_t reached = closeRuns(__ballot(valid && (seed || hit)), A);
            if(valid && ((reached >> lane) & 1ULL)) state[s + uint32_t(lane)] = uint8_t(st | (forward ? BIG_FWD : BIG_BWD));
            neighbourReached = reached; neighbourY = y;
        }
        waveLdsSync();
    }
    // ---- connected components of the active cells, 8-neighbourhood (:792-868): the smallest position of a component's cells ----
    for(int i = lane; i < n; i += WAVE) label[i] = (state[i] & (BIG_FWD | BIG_BWD)) == (BIG_FWD | BIG_BWD) ? uint16_t(i) : uint16_t(0xffff);
    waveLdsSync();
    for(int sweep = 0; sweep < 2 * CELLS_BIG_COLUMNS + 4; sweep++) {
        const bool forward = (sweep & 1) == 0;
        bool changed = false;
        uint64_t neighbourActive = 0;
        uint32_t neighbourY = 0, neighbourLabel = 0xffffu;
        for(int32_t step = 0; step <= int32_t(lastColumn) - int32_t(firstColumn); step++) {
            const uint32_t column = forward ? firstColumn + uint32_t(step) : lastColumn - uint32_t(step);
            const uint32_t s = colStart[column], m = colStart[column + 1u] - s;
            if(m == 0) { neighbourActive = 0; continue; }
            const bool valid = uint32_t(lane) < m;
            const uint32_t y = keys[s + (valid ? uint32_t(lane) : 0u)] >> 16;
            const uint32_t before = valid ? uint32_t(label[s + uint32_t(lane)]) : 0xffffu;
            const bool active = before != 0xffffu;
            uint32_t mine = before;
            for(uint64_t todo = neighbourActive; todo; todo &= todo - 1) {
                const int j = __ffsll((unsigned long long)todo) - 1;
                const uint32_t otherY = __builtin_amdgcn_readlane(neighbourY, j), otherLabel = __builtin_amdgcn_readlane(neighbourLabel, j);
                if(active && (y + 1u - otherY) <= 2u) mine = min(mine, otherLabel);
            }
            // Inside the column: along runs of consecutive iY whose cells are active.
            const uint64_t activeLanes = __ballot(active);
            const uint32_t above = uint32_t(__shfl_down(int(y), 1, WAVE));
            const uint64_t A = __ballot(valid && uint32_t(lane) + 1u < m && above == y + 1u) & activeLanes & (activeLanes >> 1);
            for(;;) {
                const uint32_t up = uint32_t(__shfl_down(int(mine), 1, WAVE)), down = uint32_t(__shfl_up(int(mine), 1, WAVE));
                uint32_t next = mine;
                if((A >> lane) & 1ULL) next = min(next, up);
                if(lane > 0 && ((A >> (lane - 1)) & 1ULL)) next = min(next, down);
                const bool moved = next != mine;
                mine = next;
                if(!__any(moved)) break;
            }
            if(active && mine != before) { label[s + uint32_t(lane)] = uint16_t(mine); changed = true; }
            neighbourActive = activeLanes; neighbourY = y; neighbourLabel = mine;
        }
        waveLdsSync();
        if(!__any(changed) && sweep >= 1) break;             // (the first backward sweep that changes nothing: a fixed point of both directions)
    }
    // ---- one banded alignment per component (:890-934) ----
    for(int i = lane; i < n; i += WAVE) {
        const uint32_t l = label[i];
        if(l == 0xffffu) continue;
        atomicMin(&yMin[l], keys[i] >> 16);
        atomicMax(&yMax[l], keys[i] >> 16);
    }
    waveLdsSync();
    for(int i = lane; i < n; i += WAVE) {
        if(uint32_t(label[i]) != uint32_t(i)) continue;
        const uint32_t YMin = yMin[i] * opt.deltaY;
        const uint32_t YMax = (yMax[i] + 1) * opt.deltaY - 1;
        const int32_t bandMin = int32_t(nx) - 1 - int32_t(YMax);
        const int32_t bandMax = int32_t(nx) - 1 - int32_t(YMin);
        const int32_t bandWidth = bandMax - bandMin + 1;
        if(int64_t(bandWidth) > int64_t(opt.maxBand)) continue;                // :929
        DpTask task; task.pair = pair; task.bandMin = bandMin; task.bandMax = bandMax; task.label = keys[i];
        if(bandWidth > 1024) {
            // (to the back of the task list, for the wide DP: see align4CellsKernel)
            const uint32_t w = atomicAdd(taskCount + CELLS_WIDE_COUNTER, 1u);
            if(w < taskCapacity) tasks[taskCapacity - 1u - w] = task;
        } else {
            const uint32_t t = atomicAdd(taskCount, 1u);
            if(t < taskCapacity) tasks[t] = task;
        }
    }
    return true;
}
// BIG (align4CellsLongBigKernel): LONG for the candidates that keep more cells than its graphs hold -- one candidate at a time, up to
// CELLS_BIG_KEPT kept cells in LDS behind the wavefronts' slots, the graph by cellsBigGraph (wavefront 0, in the LDS of the tables).
template<int Q, bool DUMP, bool LONG, bool BIG = false>
__device__ __forceinline__ void cellsChunkBody(
    const uint32_t* __restrict__ kmerIds, const PairDesc* __restrict__ pairs,
    const CellsChunk* __restrict__ chunks, uint32_t chunkCount, const uint32_t* __restrict__ members,
    const DeviceOptions& opt, uint32_t magicX, uint32_t magicY,
    DpTask* __restrict__ tasks, uint32_t* __restrict__ taskCount, uint32_t taskCapacity,
    uint8_t* __restrict__ pairFlags, uint32_t* __restrict__ activeKeys, uint32_t* __restrict__ activeCounts, const HitLists& hitLists)
{
    extern __shared__ uint32_t ldsWords[];
    __shared__ uint32_t waveTotals[(LONG ? CELLS_LONG_THREADS : SHASTA_CELLS_MAX_THREADS) / 64];
    constexpr int IY_BITS = LONG ? CELLS_LONG_IY_BITS : CELLS_IY_BITS, COUNT_BITS = LONG ? CELLS_LONG_COUNT_BITS : CELLS_COUNT_BITS;
    constexpr int MAXC = 64 * Q;
    static_assert(!BIG || (LONG && !DUMP), "the large graph belongs to the windowed class");
    constexpr uint32_t KEPT_ROOM = BIG ? uint32_t(CELLS_BIG_KEPT) : uint32_t(MAXC);          // kept cells listed per candidate
    if(blockIdx.x >= chunkCount) return;
    const CellsChunk chunk = chunks[blockIdx.x];
    const int lane = laneId();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), waves = blockDim.x >> 6;       // scalar: loop control on the scalar unit
    const uint32_t NA = 1u << chunk.naLog2, SC = 1u << chunk.scLog2;
    const int xBits = int(chunk.naLog2), scShift = 32 - int(chunk.scLog2);
    const uint32_t xMask = NA - 1;
    constexpr int F = cellsBucketFactorLog2(Q);
    const int bucketShift = 32 - (xBits + F);                     // 2^F NA buckets
    const uint32_t bucketCount = NA << F, rangeWords = bucketCount >> 1;
    uint32_t* const range = ldsWords;                             // 16-bit halves: where bucket b's entries start (its end while the table is built)
    uint32_t* const entries = range + rangeWords + CELLS_RANGE_PAD;                         // (hash32(kmer id) << xBits) | ordinal, bucket after bucket
    uint32_t* const cells = entries + NA;
    uint32_t* const slots = cells + SC;                           // slot w: kept[MAXC], later the map[2 MAXC] of the graph | scratch[8] | stage[4 CELLS_STAGE]
    // The slot the candidate being streamed appends its kept cells to (every wavefront), and this wavefront's own slot
    // (graph of the candidate it was given, staged tasks).  scratch: [0] kept count, [1] min, [2] max, [3] staged tasks,
    // [4] what went wrong while streaming.
    const uint32_t scratchAt = uint32_t(MAXC + cellsQueueWords(Q));
    uint32_t* kept = slots;
    uint32_t* scratch = kept + scratchAt;
    uint32_t* const ownKept = slots + wave * cellsSlotLdsWords(Q);
    uint32_t* const ownScratch = ownKept + scratchAt;
    uint32_t* const stage = ownScratch + 8;
    const uint32_t threshold = uint32_t(opt.minEntryCountPerCell > 1 ? min(opt.minEntryCountPerCell, uint64_t(0xffffffffu)) : 1);
    uint32_t* const bigKept = slots + waves * cellsSlotLdsWords(Q);       // (BIG: the one candidate's kept cells, behind the slots)
    PHASE_BEGIN();


    // --- table of the read shared by every candidate of the chunk: read 0, or (swapped chunk:
    //     candidates gathered by the host because they share a short read 1) read 1 ---
    const PairDesc pdFirst = pairs[members[chunk.firstMember]];
    const bool chunkSwapped = (chunk.swapped & 1) != 0, noGrid = LONG || (chunk.swapped & 2) != 0;
    // The exact bucketed table: h = kmerId * odd constant is a bijection of the 32-bit values, so (bucket = the top bits of h, the
    // other bits of h) IS the kmer id.  2^F NA buckets (F = cellsBucketFactorLog2(Q) = 1) for at most NA markers (load factor below 1/2: the loop over a bucket's
    // entries below runs as long as the fullest bucket of a round's 256 markers); an entry -- the other bits of h | the
    // ordinal -- is one word, the buckets are the segments of one array, and where they start is a table of 16-bit positions,
    // two to a word.  A counting sort: the sizes by LDS atomics on the halves, one scan that leaves every bucket's END, and the
    // fill counts each end down to the bucket's start (the reference's own way of filling its buckets,
    // MemoryMappedVectorOfVectors::storeMultithreaded) -- no compare-and-swap loops, nothing that can overflow.
    if(lane == 0) ownScratch[3] = 0;
    // tabSeq[0 .. tabCount): the tabled read, or (LONG) a window of it; tabCount < NA, or = NA for a window (the entries' ordinal
    // field holds NA values, the 16-bit bucket starts NA <= 2^13 positions).
    auto buildTable = [&](const uint32_t* __restrict__ tabSeq, const uint32_t tabCount) {
    for(uint32_t k = threadIdx.x; k < rangeWords; k += blockDim.x) range[k] = 0;
    __syncthreads();
    // (Four markers per thread and trip, loaded before any is used: one marker per trip waited for memory every trip.)
    const uint32_t tabLast = tabCount ? tabCount - 1u : 0u;
    const uint32_t* __restrict__ const tabSafe = tabCount ? tabSeq : kmerIds;
    for(uint32_t t0 = threadIdx.x; t0 < tabCount; t0 += 4u * blockDim.x) {
        uint32_t km[4];
#pragma unroll
        for(int u = 0; u < 4; u++) km[u] = tabSafe[min(t0 + uint32_t(u) * blockDim.x, tabLast)];
#pragma unroll
        for(int u = 0; u < 4; u++) {
            if(t0 + uint32_t(u) * blockDim.x >= tabCount) continue;
            const uint32_t b = hash32(km[u]) >> bucketShift;
            atomicAdd(&range[b >> 1], (b & 1u) ? 0x10000u : 1u);
        }
    }
    __syncthreads();
    {
        const uint32_t per = (rangeWords + blockDim.x - 1) / blockDim.x, first = threadIdx.x * per;
        uint32_t sum = 0;
        for(uint32_t k = first; k < min(first + per, rangeWords); k++) { const uint32_t w = range[k]; sum += (w & 0xffffu) + (w >> 16); }
        uint32_t inclusive = sum;
#pragma unroll
        for(int d = 1; d < WAVE; d <<= 1) { const uint32_t o = uint32_t(__shfl_up(int(inclusive), d, WAVE)); if(lane >= d) inclusive += o; }
        if(lane == WAVE - 1) waveTotals[wave] = inclusive;
        __syncthreads();
        uint32_t base = inclusive - sum;
        for(uint32_t w = 0; w < wave; w++) base += waveTotals[w];
        for(uint32_t k = first; k < min(first + per, rangeWords); k++) {
            const uint32_t w = range[k];
            const uint32_t endLow = base + (w & 0xffffu), endHigh = endLow + (w >> 16);
            range[k] = endLow | (endHigh << 16);
            base = endHigh;
        }
        if(threadIdx.x == 0) range[rangeWords] = tabCount;                 // starts[bucketCount]: where the last bucket ends
    }
    __syncthreads();
    for(uint32_t t0 = threadIdx.x; t0 < tabCount; t0 += 4u * blockDim.x) {
        uint32_t km[4];
#pragma unroll
        for(int u = 0; u < 4; u++) km[u] = tabSafe[min(t0 + uint32_t(u) * blockDim.x, tabLast)];
#pragma unroll
        for(int u = 0; u < 4; u++) {
            const uint32_t t = t0 + uint32_t(u) * blockDim.x;
            if(t >= tabCount) continue;
            const uint32_t h = hash32(km[u]);
            const uint32_t b = h >> bucketShift;
            const uint32_t old = atomicAdd(&range[b >> 1], (b & 1u) ? 0xffff0000u : 0xffffffffu);   // (minus one in the half: a low half never borrows, it counts down to its bucket's start)
            entries[((b & 1u) ? old >> 16 : old & 0xffffu) - 1u] = (h << (32 - bucketShift)) | t;
        }
    }
    __syncthreads();
    };
    if(!LONG) buildTable(kmerIds + (chunkSwapped ? pdFirst.begin1 : pdFirst.begin0), chunkSwapped ? pdFirst.ny : pdFirst.nx);      // (fewer than NA markers: the host's classes)
    PHASE_MARK(0);

    // The stream in groups of 64 markers dealt to the wavefronts in turn (group g to wavefront g mod waves: every wavefront
    // gets the same number of groups, give or take one); a wavefront takes CELLS_UNROLL of its groups per round, and a slot of
    // the last round whose group lies beyond the stream is skipped (wavefront-uniform).  Slot u of the round that starts at
    // s0 holds markers s0 + u groupStride + lane.
    const uint32_t groupStride = waves * uint32_t(WAVE);
    const uint32_t roundStride = uint32_t(CELLS_UNROLL) * groupStride;
    const uint32_t firstRound = wave * uint32_t(WAVE);
    // A candidate starts with a chain of dependent global loads (member list -> pair descriptor -> the first markers of
    // its stream): the descriptor is loaded one candidate ahead.  (Loading the first round of markers ahead as well cost
    // twelve vector registers and bought nothing measurable: the wait it removed was taken by the next barrier.)
    uint32_t pairAhead = members[chunk.firstMember];
    PairDesc pdAhead = pdFirst;

    // Groups of `waves` candidates: streamed one after the other by the whole workgroup, then one graph per wavefront.
    for(uint32_t group = 0; group < chunk.count; group += (BIG ? 1u : waves)) {
    const uint32_t groupEnd = min(group + (BIG ? 1u : waves), uint32_t(chunk.count));
    for(uint32_t c = group; c < groupEnd; c++) {
        const uint32_t pair = pairAhead;
        const PairDesc pd = pdAhead;
        const uint32_t nx = pd.nx, ny = pd.ny;
        const bool more = c + 1 < uint32_t(chunk.count);
        if(more) { pairAhead = members[chunk.firstMember + c + 1]; pdAhead = pairs[pairAhead]; }
        kept = slots + (c - group) * cellsSlotLdsWords(Q);
        scratch = kept + scratchAt;
        if(BIG) kept = bigKept;
        const bool swapped = LONG ? ny < nx : chunkSwapped;        // (LONG: every candidate tables its own shorter read)
        const uint32_t* __restrict__ stream = kmerIds + (swapped ? pd.begin0 : pd.begin1);
        const uint32_t streamCount = swapped ? nx : ny;
        int overflow = 0, reason = 0;
        bool gaveUp = false;                                       // wave-uniform: the candidate's cell table is full (see drain)

        // When the candidate's whole cell grid fits the wavefront's cell region as one BYTE per cell (nx + ny up to about 4000
        // at the default cell size: nearly every candidate of the first class), the entries are counted in a direct grid: one LDS
        // atomic per hit, no keys, no probing.  A lane adds only while the byte is below the threshold, so a byte stays below
        // threshold + the adds in flight between such a read and its add (at most 64 per wave instruction, four slots, four
        // wavefronts: more than a byte holds only if they all hit ONE cell, which takes a tandem repeat); the add returns the byte
        // as it was, and one that finds 255 has carried into the next byte: the candidate is flagged (reason 8) and the host runs
        // it again with bit 1 of `swapped` set -- counted in the open-addressing table of packed (iY | iX | count) words, whose
        // ten count bits hold any cell (cellsClassRule.packedOk).  That table also serves the candidates whose grid does not fit.
        const uint32_t gridX = divMagic(nx + ny - 2, magicX) + 1, gridY = divMagic(nx + ny - 2, magicY) + 1;
        const bool useGrid = !noGrid && nx + ny >= 2 && uint64_t(gridX) * gridY <= 4ull * SC && threshold <= 191;
        {
            const uint32_t first = threadIdx.x, stride = blockDim.x;
            if(useGrid) {
                const uint32_t gridWords = (gridX * gridY + 3) / 4;
                for(uint32_t k = first; k < gridWords; k += stride) cells[k] = 0;
            } else {
                for(uint32_t k = first; k < SC; k += stride) cells[k] = EMPTY32;
            }
            if(first == 0) { scratch[0] = 0; scratch[1] = 0; scratch[4] = 0; scratch[5] = pair; scratch[6] = nx; scratch[7] = ny; }   // [1]: matches listed so far (the graph uses it later); [5..7]: for the graph
        }
        // The candidate's matches are also LISTED (align4_sparse.hpp: the banded alignment from the matches inside the band):
        // every wavefront appends what it drains from its queue to the candidate's list in HBM, (x << 16) | y, in no particular
        // order; what does not fit the list's capacity is counted but not stored (the count tells).
        const bool listHits = !DUMP && hitLists.hits != nullptr;
        const uint64_t hitBegin = listHits ? hitLists.base[pair] : 0ULL;
        const uint32_t hitCapacity = listHits ? uint32_t(hitLists.base[pair + 1] - hitBegin) : 0u;
        uint32_t* __restrict__ const hitList = listHits ? hitLists.hits + hitBegin : nullptr;
        if(SHASTA_ABLATE != 3) __syncthreads();
        PHASE_MARK(1);

        // --- alignment matrix entries -> per-cell counts (createAlignmentMatrix + createCells) ---
        // Counts hits: hit[u] with table ordinal ti[u] and stream ordinal ts[u] (N = 4 in a round's first pass, 1 for the
        // further matches of a marker).
        auto countHits = [&](auto nTag, const bool* hit, const uint32_t* ti, const uint32_t* ts) {
            constexpr int N = decltype(nTag)::value;
            bool pending[N];
            uint32_t iX[N], iY[N];
#pragma unroll
            for(int u = 0; u < N; u++) {
                const uint32_t t = ts[u];
                const uint32_t x = swapped ? t : ti[u], y = swapped ? ti[u] : t;
                const uint32_t X = x + y, Y = nx + y - x - 1;                  // getXY, :171-177
                iX[u] = divMagic(X, magicX); iY[u] = divMagic(Y, magicY);
                pending[u] = hit[u];
            }
            if(useGrid) {
                // All reads, then all atomics, then the (rare) threshold crossings: the LDS operations of the N slots are
                // independent of each other, their latencies overlap.
                uint32_t word[N], shift[N], now[N], raw[N], before[N];
                bool add[N];
#pragma unroll
                for(int u = 0; u < N; u++) {
                    const uint32_t idx = iY[u] * gridX + iX[u];
                    word[u] = pending[u] ? idx >> 2 : 0u; shift[u] = 8u * (idx & 3u);
                }
                // (Unconditional reads, and the atomics' results looked at only after the last one is issued: behind `pending &&`
                // and inside the select each operation sat in its own branch with its own wait -- four LDS round trips one after
                // the other where two are needed.)
#pragma unroll
                for(int u = 0; u < N; u++) now[u] = ldsLoadNow(&cells[word[u]]);
#pragma unroll
                for(int u = 0; u < N; u++) { add[u] = pending[u] && ((now[u] >> shift[u]) & 0xffu) < threshold; raw[u] = 0; }
#pragma unroll
                for(int u = 0; u < N; u++) if(add[u]) raw[u] = atomicAdd(&cells[word[u]], 1u << shift[u]);
#pragma unroll
                for(int u = 0; u < N; u++) before[u] = add[u] ? (raw[u] >> shift[u]) & 0xffu : 0xffffu;
#pragma unroll
                for(int u = 0; u < N; u++) {
                    if(before[u] + 1 == threshold) {                                          // :417
                        const uint32_t at = atomicAdd(&scratch[0], 1u);
                        if(at < KEPT_ROOM) kept[at] = (iY[u] << 16) | iX[u];
                    }
                    if(before[u] == 255u) { overflow = max(overflow, 1); reason |= 8; }     // the byte wrapped
                }
                return;
            }
            // The LDS cell table packs (iY:12 | iX:10 | count:10) in one word; the host only sends
            // candidates whose cell indices fit (others run in the HBM-scratch kernel).
            uint32_t key[N], packed[N], cs[N], probes[N];
#pragma unroll
            for(int u = 0; u < N; u++) {
                key[u] = pending[u] ? ((iY[u] << 16) | iX[u]) : EMPTY32;
                packed[u] = (iY[u] << CELLS_IX_BITS) | iX[u];                   // (LONG: fourteen bits of iY, the host's class rule)
                cs[u] = hash32(key[u]) >> scShift;
                probes[u] = 0;
            }
            bool anyPending = false;
#pragma unroll
            for(int u = 0; u < N; u++) anyPending |= pending[u];
            while(__any(anyPending)) {
                anyPending = false;
#pragma unroll
                for(int u = 0; u < N; u++) {
                    if(pending[u]) {
                        const uint32_t cur = ldsLoadNow(&cells[cs[u]]);
                        bool done = false;
                        uint32_t before = 0;
                        if(cur != EMPTY32 && (cur >> COUNT_BITS) == packed[u]) {
                            if(LONG && (cur & ((1u << COUNT_BITS) - 1)) >= threshold) before = threshold;      // (a kept cell: its count is not needed any more)
                            else {
                                before = atomicAdd(&cells[cs[u]], 1u) & ((1u << COUNT_BITS) - 1);
                                if(LONG && before == (1u << COUNT_BITS) - 1) { overflow = max(overflow, 1); reason |= 1; }     // (carried into the key)
                            }
                            done = true;
                        } else if(cur == EMPTY32) {
                            // Claim the slot; on failure look at the same slot again.
                            done = atomicCAS(&cells[cs[u]], EMPTY32, (packed[u] << COUNT_BITS) | 1u) == EMPTY32;
                        } else {
                            cs[u] = (cs[u] + 1) & (SC - 1);
                            if(++probes[u] == SC) { overflow = max(overflow, 1); reason |= 1; pending[u] = false; }
                        }
                        if(done) {
                            if(before < threshold && before + 1 >= threshold) {                // :417
                                const uint32_t idx = atomicAdd(&scratch[0], 1u);
                                if(idx < KEPT_ROOM) kept[idx] = key[u];
                            }
                            pending[u] = false;
                        }
                    }
                    anyPending |= pending[u];
                }
            }
        };

        // LONG: the tabled read in windows of NA markers, the stream once per window; table ordinal = window's first marker + the entry's.
        const uint32_t tabTotal = LONG ? (swapped ? ny : nx) : 1u;
        const uint32_t* __restrict__ const tabAll = kmerIds + (swapped ? pd.begin1 : pd.begin0);
        for(uint32_t windowBase = 0; windowBase < tabTotal; windowBase += (LONG ? NA : 1u)) {
        if(LONG) buildTable(tabAll + windowBase, min(NA, tabTotal - windowBase));          // (ends with a barrier)
        uint32_t kmNext[CELLS_UNROLL];
        // (Unconditional loads from a clamped position: behind a branch the compiler cannot count the loads in flight and
        // waits for all of them -- vmcnt(0) right after issuing the next round's -- where the marker it needs arrived long ago.)
        const uint32_t* __restrict__ const streamSafe = streamCount ? stream : kmerIds;
        const uint32_t streamLast = streamCount ? streamCount - 1u : 0u;
#pragma unroll
        for(int u = 0; u < CELLS_UNROLL; u++) kmNext[u] = streamSafe[min(firstRound + u * groupStride + lane, streamLast)];
        SUBPHASE_DECLARE();
        // Matches wait in a queue of the wavefront (the part of its slot that the kept list does not use while candidates are
        // streamed) and are counted 128 at a time: the loop over the entries of the markers' buckets is thin -- most lanes have
        // nothing to look at in its later trips -- and only finds matches; the cell arithmetic and the LDS atomics run on
        // full wavefronts.  An entry: table ordinal << 16 | stream ordinal (both below 2^16: cellsClassFor).
        uint32_t* const queue = ownKept + MAXC;
        const uint32_t queueCapacity = uint32_t(cellsQueueWords(Q));
        uint32_t queued = 0;                                                    // wave-uniform
        // (A candidate whose cell table is full climbs to a larger one whatever else is counted -- PAIR_RESOURCE below -- and every
        // further match would walk the WHOLE table before saying so again: SC probes per drained match, 36 s per candidate of two
        // repeat-rich reads of 10 000 markers on the emulated build, real reads at k = 10.  The wavefront that has seen it, and in
        // the windowed class every wavefront of the candidate through the slot's word, stops counting and streaming.)
        auto drain = [&]() {
            if(!gaveUp && LONG && (ldsLoadNow(&scratch[4]) & 0x18u) != 0u) gaveUp = true;
            if(gaveUp) { queued = 0; return; }
            waveLdsSync();
            uint32_t listAt = 0;
            if(listHits) {
                if(lane == 0) listAt = atomicAdd(&scratch[1], queued);
                listAt = __builtin_amdgcn_readfirstlane(listAt);
            }
#pragma nounroll
            for(uint32_t base = 0; base < queued; base += uint32_t(CELLS_DRAIN) * WAVE) {   // (CELLS_DRAIN entries per lane and pass: the counting code's registers)
                bool hit[CELLS_DRAIN];
                uint32_t ti[CELLS_DRAIN], ts[CELLS_DRAIN];
#pragma unroll
                for(int k = 0; k < CELLS_DRAIN; k++) {
                    const uint32_t at = base + uint32_t(k) * WAVE + uint32_t(lane);
                    hit[k] = at < queued;
                    const uint32_t e = queue[hit[k] ? at : 0u];
                    ti[k] = (e >> 16) + (LONG ? windowBase : 0u); ts[k] = e & 0xffffu;
                    if(listHits && hit[k] && listAt + at < hitCapacity) hitList[listAt + at] = swapped ? ((ts[k] << 16) | ti[k]) : ((ti[k] << 16) | ts[k]);
                }
                if(SHASTA_ABLATE != 4) countHits(std::integral_constant<int, CELLS_DRAIN>{}, hit, ti, ts);
                if(__any(overflow != 0)) {
                    gaveUp = true;
                    if(LONG && lane == 0) atomicOr(&scratch[4], 0x08u);          // (which lane saw it and why follows at the end of the stream)
                    break;
                }
            }
            waveLdsSync();
            queued = 0;
        };
        for(uint32_t s0 = firstRound; s0 < (SHASTA_ABLATE == 2 ? 0u : streamCount); s0 += roundStride) {
            if(gaveUp) break;                                                   // (wave-uniform; the barriers are outside this loop)
            SUBPHASE_START(); SUBPHASE_COUNT(4);
            // first[u], left[u]: the entries of marker u's bucket that have not been looked at; wanted[u]: its hash bits.
            uint32_t first[CELLS_UNROLL], left[CELLS_UNROLL], wanted[CELLS_UNROLL], ts[CELLS_UNROLL];
            uint32_t most = 0;
#pragma unroll
            for(int u = 0; u < CELLS_UNROLL; u++) {
                const uint32_t km = kmNext[u];
                kmNext[u] = streamSafe[min(s0 + roundStride + u * groupStride + lane, streamLast)];      // prefetch the next round
                ts[u] = s0 + uint32_t(u) * groupStride + uint32_t(lane);
                const uint32_t h = hash32(km);
                const uint32_t b = h >> bucketShift;
                const uint16_t* const starts = reinterpret_cast<const uint16_t*>(range);
                const uint32_t begin = starts[b], end = starts[b + 1u];
                wanted[u] = h << (32 - bucketShift);
                first[u] = begin;
                left[u] = ts[u] < streamCount ? end - begin : 0u;
                most = max(most, left[u]);
            }
            SUBPHASE_ADD(0);
            uint32_t e[CELLS_UNROLL];
#pragma unroll
            for(int u = 0; u < CELLS_UNROLL; u++) e[u] = entries[left[u] != 0u ? first[u] : 0u];
            while(SHASTA_ABLATE != 5 && __any(most != 0u)) {
                SUBPHASE_COUNT(5);
                bool hit[CELLS_UNROLL];
                uint32_t packed[CELLS_UNROLL];
#pragma unroll
                for(int u = 0; u < CELLS_UNROLL; u++) {
                    const bool has = left[u] != 0u;
                    hit[u] = has && ((e[u] ^ wanted[u]) >> (32 - bucketShift)) == 0u;     // exact: the hash is a bijection, the bucket is the rest of its bits
                    packed[u] = ((e[u] & xMask) << 16) | ts[u];
                    first[u] += 1u; left[u] -= has ? 1u : 0u;
                }
                most = most != 0u ? most - 1u : 0u;
                if(__any(most != 0u)) {
#pragma unroll
                    for(int u = 0; u < CELLS_UNROLL; u++) e[u] = entries[left[u] != 0u ? first[u] : 0u];
                }
                // The matches of the four groups enter the queue together -- a ballot per group, a lane's position by the
                // popcounts below it; the queue holds a whole trip (CELLS_QUEUE), so it is emptied at ONE place, before the
                // trip that would not fit.  (Round 3's first form pushed two groups at a time through a loop that rebuilt the
                // flags with selects: 56 of a trip's 116 vector instructions.)
                uint64_t votes[CELLS_UNROLL];
                uint32_t count[CELLS_UNROLL], total = 0;
#pragma unroll
                for(int u = 0; u < CELLS_UNROLL; u++) { votes[u] = __ballot(hit[u]); count[u] = uint32_t(__popcll(votes[u])); total += count[u]; }
                if(total != 0u) {
                    if(queued + total > queueCapacity) { SUBPHASE_COUNT(6); drain(); }
                    uint32_t at = queued;
#pragma unroll
                    for(int u = 0; u < CELLS_UNROLL; u++) {
                        if(hit[u]) queue[at + uint32_t(__popcll(votes[u] & laneMaskLt()))] = packed[u];
                        at += count[u];
                    }
                    queued = at; SUBPHASE_COUNT_N(2, total);
                }
                SUBPHASE_ADD(1);
            }
        }
        if(queued) { SUBPHASE_COUNT(7); drain(); }
        SUBPHASE_FLUSH();
        if(LONG && windowBase + NA < tabTotal) __syncthreads();       // (every wavefront is done with this window's table)
        }
        // What went wrong in any wavefront's share of the rounds reaches the candidate's graph through its slot.
        if(overflow | reason) atomicOr(&scratch[4], uint32_t(reason & 7) | ((reason & 8) ? 0x20u : 0u) | (overflow == 2 ? 0x10u : (overflow == 1 ? 0x08u : 0u)));
        if(SHASTA_ABLATE != 3) __syncthreads();                       // the cell region is cleared for the next candidate
        if(listHits && threadIdx.x == 0) hitLists.meta[pair] = min(scratch[1], 0x7fffffffu) | (swapped ? 0x80000000u : 0u);
        PHASE_MARK(2);
    }

        if constexpr (BIG) {
            // The one candidate's graph: wavefront 0, working in the LDS of the tables (the others wait at the barrier below).
            if(wave == 0) {
                scratch = slots + scratchAt;
                const uint32_t pair = scratch[5], nx = scratch[6], ny = scratch[7];
                const uint32_t seen = scratch[4];
                const uint32_t n = scratch[0];
                static_assert(CELLS_BIG_WORK_WORDS <= (1 << 13) + CELLS_RANGE_PAD + (1 << 13) + (1 << 13), "the graph's arrays fit the tables' LDS (class geometry: align4.hip)");
                bool fits = seen == 0 && n <= KEPT_ROOM;
                int reasons = int(seen & 7u) | (n > KEPT_ROOM ? 2 : 0);
                if(fits && n) { fits = cellsBigGraph(bigKept, int(n), ldsWords, pair, nx, ny, opt, tasks, taskCount, taskCapacity); if(!fits) reasons |= 2; }
                if(!fits && lane == 0) pairFlags[pair] = (seen & 0x10u) ? PAIR_TOO_LONG : uint8_t(PAIR_RESOURCE | (reasons << 4));
            }
        } else
        // The kept-cell graphs of the group, one per wavefront.
        if(SHASTA_ABLATE != 1 && group + wave < groupEnd) do {
        kept = ownKept; scratch = ownScratch;
        const uint32_t pair = scratch[5], nx = scratch[6], ny = scratch[7];
        const uint32_t seen = scratch[4];
        int overflow = (seen & 0x10u) ? 2 : ((seen & 0x08u) ? 1 : 0), reason = int(seen & 7u) | ((seen & 0x20u) ? 8 : 0);
        const int n = int(scratch[0]);
        if(n > MAXC) { overflow = max(overflow, 1); reason |= 2; }
        const uint64_t anyHard = __ballot(overflow == 2), anySoft = __ballot(overflow == 1);
        if(DUMP && (anyHard || anySoft)) { if(lane == 0) activeCounts[blockIdx.x] = 0xffffffffu; break; }
        if(anyHard || anySoft) {
            // Bits 4-7 carry the reason: cell table full / kept list full / geometry (diagnostics) and, bit 7, a byte of the grid
            // that overflowed (the host runs the candidate again in the packed table of the same class) or, from the table build, a
            // tabled read whose markers crowd their buckets (the whole chunk climbs a class).
            const int reasons = (__ballot(reason & 1) ? 1 : 0) | (__ballot(reason & 2) ? 2 : 0) | (__ballot(reason & 4) ? 4 : 0) | (__ballot(reason & 8) ? 8 : 0);
            if(lane == 0) pairFlags[pair] = anyHard ? PAIR_TOO_LONG : uint8_t(PAIR_RESOURCE | (reasons << 4));
            break;
        }
        if(n == 0) { if(DUMP && lane == 0) activeCounts[blockIdx.x] = 0; break; }
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
        // The kept cells go into a small open-addressing map (cell -> index in the list) that takes the place of the list
        // in the slot -- the keys are in registers by now -- and each lane looks its eight neighbours up: a fixed amount of work
        // per candidate.  (Until round 2 every lane compared its cells with every kept cell, one readlane at a time: with the
        // stream four times faster that loop had become the longest part of a candidate with an alignment.)
        uint64_t before[Q][Q], after[Q][Q];
#pragma unroll
        for(int q = 0; q < Q; q++)
#pragma unroll
            for(int r = 0; r < Q; r++) { before[q][r] = 0; after[q][r] = 0; }
        {
            constexpr int MAP = 2 * MAXC, MAP_LOG2 = (Q == 2 ? 8 : 9), IDX_BITS = MAP_LOG2 - 1;
            // (LONG: 24 + 8 bits fill the word; the all-ones entry is no cell: the class rule keeps iY below 2^14 - 1)
            static_assert(MAP == (1 << MAP_LOG2) && CELLS_IX_BITS + IY_BITS + IDX_BITS <= (LONG ? 32 : 31), "map entry");
            uint32_t* const map = kept;
            waveLdsSync();                                                  // every lane has read its keys
            for(int k = lane; k < MAP; k += WAVE) map[k] = EMPTY32;
            waveLdsSync();
#pragma unroll
            for(int q = 0; q < Q; q++) {
                if(key[q] == EMPTY32) continue;
                const uint32_t packed = ((key[q] >> 16) << CELLS_IX_BITS) | (key[q] & 0xffffu);
                const uint32_t entry = (packed << IDX_BITS) | uint32_t(lane + q * WAVE);
                uint32_t h = hash32(packed) >> (32 - MAP_LOG2);
                while(atomicCAS(&map[h], EMPTY32, entry) != EMPTY32) h = (h + 1) & uint32_t(MAP - 1);
            }
            waveLdsSync();
#pragma unroll
            for(int q = 0; q < Q; q++) {
                if(q >= nq) break;
                const int32_t iX = int32_t(key[q] & 0xffffu), iY = int32_t(key[q] >> 16);
                uint32_t target[8], h[8], e[8];
                bool live[8];
#pragma unroll
                for(int d = 0; d < 8; d++) {                                 // the eight neighbours: all reads first
                    const int dX = (d < 3) ? -1 : (d < 5 ? 0 : 1), dY = (d < 3) ? d - 1 : (d == 3 ? -1 : (d == 4 ? 1 : d - 6));
                    const int32_t nX = iX + dX, nY = iY + dY;
                    live[d] = key[q] != EMPTY32 && uint32_t(nX) < (1u << CELLS_IX_BITS) && uint32_t(nY) < (1u << IY_BITS) - (LONG ? 1u : 0u);
                    target[d] = (uint32_t(nY) << CELLS_IX_BITS) | uint32_t(nX);
                    h[d] = hash32(target[d]) >> (32 - MAP_LOG2);
                    e[d] = map[live[d] ? h[d] : 0u];
                }
#pragma unroll
                for(int d = 0; d < 8; d++) {
                    const int dX = (d < 3) ? -1 : (d < 5 ? 0 : 1);
                    while(live[d] && e[d] != EMPTY32 && (e[d] >> IDX_BITS) != target[d]) { h[d] = (h[d] + 1) & uint32_t(MAP - 1); e[d] = map[h[d]]; }
                    const bool found = live[d] && e[d] != EMPTY32;
                    const uint32_t index = e[d] & uint32_t(MAXC - 1);
                    const uint64_t bit = 1ULL << (index & 63u);
#pragma unroll
                    for(int r = 0; r < Q; r++) {
                        if(found && int(index >> 6) == r) {
                            if(dX <= 0) before[q][r] |= bit;
                            if(dX >= 0) after[q][r] |= bit;
                        }
                    }
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
                // (cells beyond n hold no key: empty masks; a `break` here makes the index dynamic and the masks go to scratch memory)
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
                // (cells beyond n hold no key: empty masks; a `break` here makes the index dynamic and the masks go to scratch memory)
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
        if(DUMP) {
            uint32_t base = 0;
#pragma unroll
            for(int q = 0; q < Q; q++) {
                if((remaining[q] >> lane) & 1ULL) activeKeys[size_t(blockIdx.x) * MAXC + base + uint32_t(__popcll(remaining[q] & laneMaskLt()))] = key[q];
                base += uint32_t(__popcll(remaining[q]));
            }
            if(lane == 0) activeCounts[blockIdx.x] = base;
            break;
        }
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
                    // (cells beyond n hold no key: empty masks; a `break` here makes the index dynamic and the masks go to scratch memory)
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
                if(bandWidth > 1024) {
                    // (to the back of the task list, for the wide DP: see align4CellsKernel)
                    if(lane == 0) {
                        const uint32_t w = atomicAdd(taskCount + CELLS_WIDE_COUNTER, 1u);
                        if(w < taskCapacity) { DpTask task; task.pair = pair; task.bandMin = bandMin; task.bandMax = bandMax; task.label = seedKey; tasks[taskCapacity - 1u - w] = task; }
                    }
                }
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
        } while(false);
        __syncthreads();                                              // the slots are free for the next group
    }
    // Append this wave's staged tasks.
    waveLdsSync();
    scratch = ownScratch;
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

template<int Q, bool DUMP = false>
__global__ void __launch_bounds__(SHASTA_CELLS_MAX_THREADS) SHASTA_CELLS_OCCUPANCY
align4CellsChunkKernel(
    const uint32_t* __restrict__ kmerIds, const PairDesc* __restrict__ pairs,
    const CellsChunk* __restrict__ chunks, uint32_t chunkCount, const uint32_t* __restrict__ members,
    DeviceOptions opt, uint32_t magicX, uint32_t magicY,
    DpTask* __restrict__ tasks, uint32_t* __restrict__ taskCount, uint32_t taskCapacity,
    uint8_t* __restrict__ pairFlags, uint32_t* __restrict__ activeKeys, uint32_t* __restrict__ activeCounts, HitLists hitLists)
{
    cellsChunkBody<Q, DUMP, false>(kmerIds, pairs, chunks, chunkCount, members, opt, magicX, magicY, tasks, taskCount, taskCapacity, pairFlags, activeKeys, activeCounts, hitLists);
}

// The windowed class: sixteen wavefronts a workgroup (one workgroup per CU by its LDS: four wavefronts per SIMD stream a candidate
// together), 256 kept cells per candidate.
// ... and its candidates with more kept cells than that: one at a time, up to CELLS_BIG_KEPT kept cells (cellsBigGraph).
__global__ void __launch_bounds__(CELLS_LONG_THREADS)
align4CellsLongBigKernel(
    const uint32_t* __restrict__ kmerIds, const PairDesc* __restrict__ pairs,
    const CellsChunk* __restrict__ chunks, uint32_t chunkCount, const uint32_t* __restrict__ members,
    DeviceOptions opt, uint32_t magicX, uint32_t magicY,
    DpTask* __restrict__ tasks, uint32_t* __restrict__ taskCount, uint32_t taskCapacity,
    uint8_t* __restrict__ pairFlags, HitLists hitLists)
{
    cellsChunkBody<4, false, true, true>(kmerIds, pairs, chunks, chunkCount, members, opt, magicX, magicY, tasks, taskCount, taskCapacity, pairFlags, nullptr, nullptr, hitLists);
}

template<bool DUMP = false>
__global__ void __launch_bounds__(CELLS_LONG_THREADS)
align4CellsLongKernel(
    const uint32_t* __restrict__ kmerIds, const PairDesc* __restrict__ pairs,
    const CellsChunk* __restrict__ chunks, uint32_t chunkCount, const uint32_t* __restrict__ members,
    DeviceOptions opt, uint32_t magicX, uint32_t magicY,
    DpTask* __restrict__ tasks, uint32_t* __restrict__ taskCount, uint32_t taskCapacity,
    uint8_t* __restrict__ pairFlags, uint32_t* __restrict__ activeKeys, uint32_t* __restrict__ activeCounts, HitLists hitLists)
{
    cellsChunkBody<4, DUMP, true>(kmerIds, pairs, chunks, chunkCount, members, opt, magicX, magicY, tasks, taskCount, taskCapacity, pairFlags, activeKeys, activeCounts, hitLists);
}
