// Align4 on MI355X, K11: best component per candidate, AlignmentInfo and filters, streak compression of the stored
// alignments (/root/reference/src/Align4.cpp:126-147, 944-981; src/AssemblerAlign.cpp:439-472; src/Alignment.cpp:67-113;
// src/compressAlignment.cpp:11-67).  Included by align4.hip inside its anonymous namespace.
#pragma once

__global__ void __launch_bounds__(256)
winnerKernel(const DpTask* __restrict__ tasks, const DpResult* __restrict__ results, uint32_t taskCount,
    const unsigned long long* __restrict__ pairBest, uint32_t* __restrict__ pairWinner, uint8_t* __restrict__ pairTie, uint32_t* __restrict__ tieCounter)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= taskCount) return;
    const DpResult r = results[t];
    if(!r.passes) return;
    const DpTask task = tasks[t];
    const unsigned long long best = pairBest[task.pair];
    const unsigned long long key = ((unsigned long long)r.markerCount << 32) | (unsigned long long)(0xffffffffu - task.label);
    if(key == best) pairWinner[task.pair] = t;
    else if((key >> 32) == (best >> 32)) { pairTie[task.pair] = 1; atomicAdd(tieCounter, 1u); }      // (rare: see resolveComponentTies)
}

// Per candidate: AlignmentInfo (src/Alignment.cpp:67-113) and the outer filters of
// src/AssemblerAlign.cpp:439-472.
__global__ void __launch_bounds__(256)
finalizeKernel(const PairDesc* __restrict__ pairs, const shasta_oriented_read_pair* __restrict__ candidates, uint32_t pairCount,
    const DpResult* __restrict__ results, const unsigned long long* __restrict__ pairBest,
    const uint32_t* __restrict__ pairWinner, const uint8_t* __restrict__ pairTie, const uint8_t* __restrict__ pairFlags,
    DeviceOptions opt, int wantOrdinals,
    uint8_t* __restrict__ status, shasta_alignment_data* __restrict__ rows,
    uint32_t* __restrict__ storedFlags, uint64_t* __restrict__ ordCounts, uint64_t* __restrict__ compressedSizes)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if(p == pairCount) { storedFlags[p] = 0; ordCounts[p] = 0; compressedSizes[p] = 0; return; }
    if(p > pairCount) return;
    uint8_t st;
    uint32_t stored = 0;
    uint64_t ordCount = 0, compressedSize = 0;
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
        compressedSize = good ? r.compressedBytes : 0;               // (counted by dpMetricsKernel in its pass over the pairs)
    }
    status[p] = st;
    storedFlags[p] = stored;
    ordCounts[p] = ordCount;
    compressedSizes[p] = compressedSize;
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
// (StreakRecord, makeStreakRecord: align4_dp.hpp -- the chain kernel counts the bytes of its tasks' alignments too)
// (writeStreakRecord too: the wave kernel writes the streaks of its tasks' alignments as it walks their chains)

// One wavefront per stored alignment: lanes flag the streak starts of 64 marker pairs at a time;
// a start lane knows its skips at once and its length when the next start is seen (the last
// start of a chunk is carried to the next chunk).  WRITE=false only counts the bytes.
// METRICS: the same pass also takes what AlignmentInfo needs from the pairs (src/Alignment.cpp:67-113, :4-31) -- offsets
// x - y and, between consecutive pairs, skips and drifts -- as per-lane partial results (the caller reduces them).
struct PairMetrics { int32_t minOffset = 0x7fffffff, maxOffset = int32_t(0x80000000); long long sumOffset = 0; uint32_t maxSkip = 0, maxDrift = 0; };
template<bool WRITE, bool METRICS = false>
__device__ __forceinline__ uint64_t compressAlignmentWave(const uint32_t* __restrict__ ord, uint32_t n, uint8_t* __restrict__ out, PairMetrics* metrics = nullptr)
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
        if constexpr (METRICS) {
            if(valid) {
                const int32_t offset = int32_t(xy.x) - int32_t(xy.y);
                metrics->minOffset = min(metrics->minOffset, offset); metrics->maxOffset = max(metrics->maxOffset, offset);
                metrics->sumOffset += offset;
                if(i > 0) {
                    metrics->maxSkip = max(metrics->maxSkip, max(uint32_t(skip0), uint32_t(skip1)));
                    const int32_t drift = skip0 - skip1;                 // (x - y) of this pair minus (x - y) of the one before
                    metrics->maxDrift = max(metrics->maxDrift, uint32_t(drift < 0 ? -drift : drift));
                }
            }
        }
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
compressWriteKernel(const uint32_t* __restrict__ storedFlags, const uint32_t* __restrict__ storedIndex,
    const DpResult* __restrict__ results, const uint32_t* __restrict__ pairWinner, const uint32_t* __restrict__ ordScratch,
    uint32_t pairCount, const uint64_t* __restrict__ byteOffsets, uint8_t* __restrict__ bytes,
    uint64_t* __restrict__ compressedToc, const shasta_alignment_data* __restrict__ rows, shasta_alignment_data* __restrict__ rowsOut,
    const uint8_t* __restrict__ sparseState, uint32_t sparseStateCount, const PairDesc* __restrict__ pairs, const uint64_t* __restrict__ ordOffsets, const uint32_t* __restrict__ sorted)
{
    const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if(p >= pairCount || !storedFlags[p]) return;
    const uint32_t t = pairWinner[p];
    const DpResult r = results[t];
    if(sparseState && t < sparseStateCount && sparseState[t] == SPARSE_COMPLETE_STREAM) {
        // The wave kernel wrote the streaks while it walked the chain: they end where the task's room in the list of sorted hits ends.
        const PairDesc pd = pairs[p];
        const uint8_t* __restrict__ const from = reinterpret_cast<const uint8_t*>(sorted + sparseListBase(ordOffsets, t) + sparseListCapacity(pd.nx, pd.ny)) - r.compressedBytes;
        uint8_t* __restrict__ const to = bytes + byteOffsets[p];
        for(uint32_t i = uint32_t(laneId()); i < r.compressedBytes; i += WAVE) to[i] = from[i];
    } else
    (void)compressAlignmentWave<true>(ordScratch + 2 * r.ordBegin, r.markerCount, bytes + byteOffsets[p]);
    const uint32_t k = storedIndex[p];
    if(laneId() == 0) compressedToc[k] = byteOffsets[p];
    // The 64-byte AlignmentData row: one dword per lane.
    if(laneId() < 16) reinterpret_cast<uint32_t*>(rowsOut + k)[laneId()] = reinterpret_cast<const uint32_t*>(rows + p)[laneId()];
}

// AlignmentInfo's metrics of every task from its stored pairs (src/Alignment.cpp:67-113, :4-31), the inner acceptance
// (src/Align4.cpp:944-981) and the candidate's best component (:132-139) -- and, in the same pass over the pairs, the size of
// the task's alignment in shasta::compress form (what compressWriteKernel will write if the task becomes its candidate's stored
// alignment: until round 3 a kernel of its own read the winners' pairs a second time for that).  One wavefront per task: the
// pairs of a task are contiguous and ascending, 8 bytes per lane per round.
__global__ void __launch_bounds__(256)
dpMetricsKernel(
    const PairDesc* __restrict__ pairs, const DpTask* __restrict__ tasks, uint32_t taskCount,
    const uint32_t* __restrict__ ordScratch, DpResult* __restrict__ results, DeviceOptions opt, unsigned long long* __restrict__ pairBest,
    const uint8_t* __restrict__ sparseState, uint32_t sparseStateCount)       // (null: every task's metrics are taken here)
{
    const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if(t >= taskCount) return;
    if(sparseState && t < sparseStateCount && (sparseState[t] == SPARSE_COMPLETE || sparseState[t] == SPARSE_COMPLETE_STREAM)) return;       // sparseChainKernel took them while it walked the chain
    const int lane = laneId();
    DpResult r = results[t];
    const uint32_t count = r.markerCount;
    const uint2* __restrict__ p = reinterpret_cast<const uint2*>(ordScratch + 2 * r.ordBegin);
    PairMetrics m;
    const uint64_t compressedBytes = compressAlignmentWave<false, true>(ordScratch + 2 * r.ordBegin, count, nullptr, &m);
    int32_t minOffset = m.minOffset, maxOffset = m.maxOffset;
    long long sumOffset = m.sumOffset;
    uint32_t maxSkip = m.maxSkip, maxDrift = m.maxDrift;
#pragma unroll
    for(int d = 32; d >= 1; d >>= 1) {
        minOffset = min(minOffset, __shfl_xor(minOffset, d, WAVE)); maxOffset = max(maxOffset, __shfl_xor(maxOffset, d, WAVE));
        sumOffset += __shfl_xor(sumOffset, d, WAVE);
        maxSkip = max(maxSkip, uint32_t(__shfl_xor(int(maxSkip), d, WAVE))); maxDrift = max(maxDrift, uint32_t(__shfl_xor(int(maxDrift), d, WAVE)));
    }
    if(lane != 0) return;
    const DpTask task = tasks[t];
    const PairDesc pd = pairs[task.pair];
    if(count) { const uint2 f = p[0], l = p[count - 1]; r.first0 = f.x; r.first1 = f.y; r.last0 = l.x; r.last1 = l.y; }
    r.minOffset = minOffset; r.maxOffset = maxOffset; r.sumOffset = sumOffset; r.maxSkip = maxSkip; r.maxDrift = maxDrift;
    r.compressedBytes = uint32_t(compressedBytes < 0xffffffffULL ? compressedBytes : 0xffffffffULL);
    taskAcceptance(r, pd, task, opt, pairBest);
    results[t] = r;
}
