// Align method 3 (Shasta's default alignment method) on MI355X: the device pieces that are its own.
// Included by align4.hip inside its anonymous namespace, after the banded DP kernels it reuses.
//
// Assembler::alignOrientedReads3 (/root/reference/src/AssemblerAlign3.cpp:22-314) aligns a pair in
// two steps:
//   step 1  markers whose k-mer hash is under a threshold (down-sampling, :66-82) are aligned
//           without a band (:118-130); the aligned columns with equal k-mer ids give a range of
//           ordinal offsets (:198-223);
//   step 2  all markers are aligned inside that range +- bandExtend (:224-260), unless it is
//           wider than maxBand (:236-241).
// Both steps are the overlap DP of K10 (bandedDpForwardKernel; step 1 runs it with every
// diagonal of the small down-sampled matrix inside the band).  What is new here:
//   downsampleKernel   the kept markers of every oriented read, once per context: their kmer ids
//                      and ordinals, CSR by oriented read (a wavefront per read, ballot compaction);
//   align3BandKernel   the traceback of step 1: one lane per pair walks the packed trace, maps the
//                      matched columns back to ordinals, and appends the step-2 DP task;
//   align3WideDpKernel step 1 of the pairs whose down-sampled matrix has more than the 1024
//                      diagonals a register-resident DP task holds (long reads): one wavefront per
//                      pair, the three live anti-diagonals in LDS, 64 diagonals at a time.
// Integer work, HBM-bound only in downsampleKernel (4 bytes read per marker, 8 written per kept one).

// KmerInfo::hash (src/AssemblerKmers.cpp:182-186): MurmurHash2 (32-bit, seed 13477,
// src/MurmurHash2.cpp:37-85) of the 8 bytes of kmerId + reverseComplementKmerId(kmerId)
// (primitives.hpp).
__device__ __forceinline__ uint32_t kmerDownsamplingHash(uint32_t kmerId, uint32_t k)
{
    const uint64_t n = uint64_t(kmerId) + uint64_t(reverseComplementKmerId(kmerId, k));
    const uint32_t m = 0x5bd1e995u;
    uint32_t h = 13477u ^ 8u;
    uint32_t w = uint32_t(n);
    w *= m; w ^= w >> 24; w *= m;
    h *= m; h ^= w;
    w = uint32_t(n >> 32);
    w *= m; w ^= w >> 24; w *= m;
    h *= m; h ^= w;
    h ^= h >> 13; h *= m; h ^= h >> 15;
    return h;
}

// One wavefront per oriented read.  WRITE = false counts the kept markers (counts[2R] = 0 closes
// the array for the scan); WRITE = true writes them at dsToc[read], in ordinal order.
template<bool WRITE>
__global__ void __launch_bounds__(256)
downsampleKernel(const uint32_t* __restrict__ kmerIds, const uint64_t* __restrict__ toc, uint64_t orientedReadCount,
    uint32_t k, uint32_t hashThreshold, uint64_t* __restrict__ counts,
    const uint64_t* __restrict__ dsToc, uint32_t* __restrict__ dsKmerIds, uint32_t* __restrict__ dsOrdinals)
{
    const uint64_t r = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int lane = laneId();
    if(r > orientedReadCount) return;
    if(r == orientedReadCount) {
        if(!WRITE && lane == 0) counts[r] = 0;
        return;
    }
    const uint64_t begin = toc[r], n = toc[r + 1] - begin;
    const uint64_t outBase = WRITE ? dsToc[r] : 0;
    uint64_t kept = 0;                                   // wave-uniform
    for(uint64_t base = 0; base < n; base += WAVE) {
        const uint64_t i = base + uint64_t(lane);
        uint32_t id = 0;
        bool keep = false;
        if(i < n) {
            id = kmerIds[begin + i];
            keep = kmerDownsamplingHash(id, k) < hashThreshold;
        }
        const uint64_t votes = __ballot(keep);
        if(WRITE && keep) {
            const uint64_t pos = outBase + kept + uint64_t(__popcll(votes & laneMaskLt()));
            dsKmerIds[pos] = id;
            dsOrdinals[pos] = uint32_t(i);
        }
        kept += uint64_t(__popcll(votes));
    }
    if(!WRITE && lane == 0) counts[r] = kept;
}

// A DP over more than 1024 diagonals (step 1 of align method 3 on a long pair: every diagonal of the down-sampled matrix; a
// wide component of Align4): the same recurrence, tie policy and end-cell rule as bandedDpForwardKernel over the diagonals
// d = i - j = dMin + b, b in [0, width).  One wavefront per task sweeps the anti-diagonals s = i + j; the values of
// anti-diagonals s, s-1, s-2 live in three LDS rows of W = nx + ny + 1 words (cells that do not
// exist hold NEG_SCORE), a lane owns diagonal 64 q + lane of chunk q.  Trace: two bit planes per
// (s, q), at words 2 (s Q + q) and 2 (s Q + q) + 1 of the task's trace, Q = ceil(W / 64); codes as
// in bandedDpForwardKernel.  Sized for the few long pairs of a batch, not tuned further.
// dMin, width: the diagonals d = i - j of the task, dMin .. dMin + width - 1 (the whole matrix for step 1 of align method 3: -ny ..
// nx; the band of a component of more than 1024 diagonals for Align4, whose maxBand the reference does not bound, src/Align4.cpp:929).
struct WideTask { uint32_t pair, chunks; uint64_t traceOffset; int32_t dMin; uint32_t width; };
struct WideEnd { int32_t bestI, bestJ, score, pad; };
constexpr uint32_t ALIGN3_WIDE_MAX_DIAGONALS = 8192;      // 3 rows of 32 KB in LDS
// Beyond that (down-sampled reads of more than 4096 markers each: reads of several hundred kilobases) the three rows live
// in HBM scratch and a workgroup of four wavefronts shares the chunks of an anti-diagonal (HUGE): the reference has no limit
// here (src/AssemblerAlign3.cpp:22-314), this path's limit is the size of the trace, 2 W ceil(W / 64) words per pair.
constexpr uint32_t ALIGN3_HUGE_MAX_DIAGONALS = 65536;

template<bool HUGE, int TIE>
__global__ void __launch_bounds__(HUGE ? 256 : 64)
align3WideDpKernel(const uint32_t* __restrict__ kmerIds, const PairDesc* __restrict__ dsPairs,
    const WideTask* __restrict__ tasks, uint32_t taskCount, uint32_t rowWords,
    uint64_t* __restrict__ trace, WideEnd* __restrict__ ends, int32_t* __restrict__ hugeRows, DpScores scores)
{
    extern __shared__ int32_t wideRows[];                  // 3 x rowWords (not HUGE)
    __shared__ int32_t sBest[3 * 4];
    const uint32_t t = blockIdx.x;
    if(t >= taskCount) return;
    const int lane = laneId();
    const uint32_t wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    int32_t* const rows = HUGE ? hugeRows + size_t(t) * 3u * rowWords : wideRows;
    const WideTask task = tasks[t];
    const PairDesc pd = dsPairs[task.pair];
    const int32_t nx = int32_t(pd.nx), ny = int32_t(pd.ny);
    const uint32_t* __restrict__ p0 = kmerIds + pd.begin0;
    const uint32_t* __restrict__ p1 = kmerIds + pd.begin1;
    const int32_t W = int32_t(task.width), dMin = task.dMin;
    const uint32_t Q = task.chunks;
    uint64_t* __restrict__ tr = trace + task.traceOffset;
    for(uint32_t k = threadIdx.x; k < 3u * rowWords; k += blockDim.x) rows[k] = NEG_SCORE;
    __syncthreads();
    int32_t bestScore = NEG_SCORE, bestI = DpTie<TIE>::noEndCell, bestJ = DpTie<TIE>::noEndCell;
    for(int32_t s = 0; s <= nx + ny; s++) {
        int32_t* const cur = rows + uint32_t(s % 3) * rowWords;
        const int32_t* const prev1 = rows + uint32_t((s + 2) % 3) * rowWords;     // s - 1
        const int32_t* const prev2 = rows + uint32_t((s + 1) % 3) * rowWords;     // s - 2
        for(uint32_t q = wave; q < Q; q += waves) {
            const int32_t b = int32_t(q * 64u) + lane, d = b + dMin;
            // Cell (i, j) of this anti-diagonal on diagonal d, if it exists.
            const int32_t i2 = s + d, j2 = s - d;
            const bool exists = b < W && i2 >= 0 && j2 >= 0 && ((i2 & 1) == 0) && (i2 >> 1) <= nx && (j2 >> 1) <= ny;
            const int32_t i = i2 >> 1, j = j2 >> 1;
            int32_t v = NEG_SCORE;
            bool isV = false, isH = false, eq = false;
            if(exists) {
                if(i == 0 || j == 0) v = 0;                                            // free leading gaps
                else {
                    eq = p0[i - 1] == p1[j - 1];
                    // (a neighbour that does not exist stays far below every real score: NEG_SCORE + |gap| never beats one)
                    const int32_t dg = prev2[b] + (eq ? scores.match : scores.mismatch);
                    const int32_t vg = (b + 1 < W ? prev1[b + 1] : NEG_SCORE) + scores.gap;     // from (i, j-1)
                    const int32_t hg = (b >= 1 ? prev1[b - 1] : NEG_SCORE) + scores.gap;        // from (i-1, j)
                    // The first of the tie policy's order of (diagonal, vertical, horizontal) that reaches the maximum.
                    v = max(max(dg, vg), hg);
                    const int32_t candidates[3] = {dg, vg, hg};
                    const bool first = v == candidates[DpTie<TIE>::first], second = !first && v == candidates[DpTie<TIE>::second], third = !first && !second;
                    isV = DpTie<TIE>::first == DpTie<TIE>::VERTICAL ? first : (DpTie<TIE>::second == DpTie<TIE>::VERTICAL ? second : third);
                    isH = DpTie<TIE>::first == DpTie<TIE>::HORIZONTAL ? first : (DpTie<TIE>::second == DpTie<TIE>::HORIZONTAL ? second : third);
                }
                // Free trailing gaps: the best cell of the last row and column, ties by the policy.
                if((i == nx || j == ny) && (v > bestScore || (v == bestScore && DpTie<TIE>::endCellWins(i, j, bestI, bestJ)))) {
                    bestScore = v; bestI = i; bestJ = j;
                }
            }
            if(b < int32_t(rowWords)) cur[b] = v;
            const uint64_t loPlane = __ballot(exists && (isH || (!isV && !eq)));
            const uint64_t hiPlane = __ballot(exists && (isV || isH));
            if(lane == 0) { tr[2ULL * (uint64_t(s) * Q + q)] = loPlane; tr[2ULL * (uint64_t(s) * Q + q) + 1] = hiPlane; }
        }
        __syncthreads();
    }
#pragma unroll
    for(int dlt = 32; dlt >= 1; dlt >>= 1) {
        const int32_t os = __shfl_xor(bestScore, dlt, WAVE);
        const int32_t oi = __shfl_xor(bestI, dlt, WAVE);
        const int32_t oj = __shfl_xor(bestJ, dlt, WAVE);
        if(os > bestScore || (os == bestScore && DpTie<TIE>::endCellWins(oi, oj, bestI, bestJ))) { bestScore = os; bestI = oi; bestJ = oj; }
    }
    if(HUGE) {
        // The wavefronts' best cells meet in LDS.
        if(lane == 0) { sBest[3 * wave] = bestScore; sBest[3 * wave + 1] = bestI; sBest[3 * wave + 2] = bestJ; }
        __syncthreads();
        for(uint32_t w = 0; w < waves; w++) {
            const int32_t os = sBest[3 * w], oi = sBest[3 * w + 1], oj = sBest[3 * w + 2];
            if(os > bestScore || (os == bestScore && DpTie<TIE>::endCellWins(oi, oj, bestI, bestJ))) { bestScore = os; bestI = oi; bestJ = oj; }
        }
    }
    if(threadIdx.x == 0) { WideEnd e; e.bestI = bestI; e.bestJ = bestJ; e.score = bestScore; e.pad = 0; ends[t] = e; }
}

// Traceback of step 1, one lane per task (= per candidate that has a step-1 DP).  The trace is the
// one bandedDpForwardKernel (WIDE = false) or align3WideDpKernel (WIDE = true) wrote for the
// down-sampled pair; code 0 is a diagonal step over equal k-mer ids, the only columns that count
// (:207-216).  A pair with such columns and a band no wider than maxBand gets its step-2 task
// appended to tasks2 (order is irrelevant: one task per pair).  The band is clipped to the
// diagonals of the matrix, -ny .. nx (the reference leaves that to SeqAn, :226-227).
template<bool WIDE>
__global__ void __launch_bounds__(256)
align3BandKernel(
    const PairDesc* __restrict__ dsPairs, const PairDesc* __restrict__ pairs, const DpTask* __restrict__ tasks1,
    const uint32_t* __restrict__ sortedIds, uint32_t taskCount,
    const DpEnd* __restrict__ ends, const WideTask* __restrict__ wideTasks, const WideEnd* __restrict__ wideEnds,
    const uint64_t* __restrict__ trace, const uint32_t* __restrict__ dsOrdinals,
    int32_t bandExtend, int32_t maxBand, DpTask* __restrict__ tasks2, uint32_t* __restrict__ taskCount2, uint32_t taskCapacity2, uint32_t wideCounter)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if(idx >= taskCount) return;
    uint32_t pair;
    int32_t i, j, score, bandMin1 = 0, s0 = 0, dMinWide = 0;
    uint32_t C = 1, laneBase = 0, Q = 0;
    const uint64_t* __restrict__ tr;
    if(WIDE) {
        const WideTask task = wideTasks[idx];
        const WideEnd e = wideEnds[idx];
        pair = task.pair; i = e.bestI; j = e.bestJ; score = e.score; Q = task.chunks; dMinWide = task.dMin;
        tr = trace + task.traceOffset;
    } else {
        const uint32_t t = sortedIds[idx];
        const DpTask task = tasks1[t];
        const DpEnd e = ends[t];
        pair = task.pair; i = e.bestI; j = e.bestJ; score = e.score; bandMin1 = task.bandMin; laneBase = e.laneBase;
        const PairDesc dsGeo = dsPairs[pair];
        const DpGeometry geo = dpGeometry(task.bandMin, task.bandMax, dsGeo.nx, dsGeo.ny);
        C = uint32_t(dpDiagonals(geo.cls)); s0 = geo.s0;
        tr = trace + e.traceOffset;
    }
    if(score <= NEG_SCORE) return;
    const PairDesc ds = dsPairs[pair];
    const uint32_t RW = 2u * C;
    const uint32_t* __restrict__ ord0 = dsOrdinals + ds.begin0;
    const uint32_t* __restrict__ ord1 = dsOrdinals + ds.begin1;
    int32_t offsetMin = 0x7fffffff, offsetMax = int32_t(0x80000000);
    while(i > 0 && j > 0) {
        uint64_t lo, hi;
        uint32_t bit;
        if(WIDE) {
            const uint32_t b = uint32_t(i - j - dMinWide);
            const uint64_t w = 2ULL * (uint64_t(uint32_t(i + j)) * Q + (b >> 6));
            lo = tr[w]; hi = tr[w + 1]; bit = b & 63u;
        } else {
            const uint32_t b = uint32_t(i - j - bandMin1);
            const uint64_t it = uint64_t(uint32_t(i + j - s0) >> 1);
            const uint32_t c = b % C;
            lo = tr[it * RW + 2u * c]; hi = tr[it * RW + 2u * c + 1u]; bit = laneBase + b / C;
        }
        const uint32_t dir = uint32_t((lo >> bit) & 1ULL) | (uint32_t((hi >> bit) & 1ULL) << 1);
        if(dir == 0u) {
            const int32_t offset = int32_t(ord0[i - 1]) - int32_t(ord1[j - 1]);
            offsetMin = min(offsetMin, offset);
            offsetMax = max(offsetMax, offset);
            --i; --j;
        } else if(dir == 1u) { --i; --j; }
        else if(dir == 2u) { --j; }
        else { --i; }
    }
    // No column of the down-sampled alignment pairs equal kmer ids.  When the down-sampled alignment has no aligned column at
    // all the reference returns an empty alignment (:185-192).  When it has aligned columns but all of them are mismatches,
    // the reference goes on with offsetMin = INT_MAX, offsetMax = INT_MIN (:196-222), overflows the band arithmetic and hands
    // SeqAn a band whose lower diagonal lies above its upper one; what SeqAn 2.4.0 does with it cannot be checked here (the
    // library is absent), the caller logs and skips such a pair if it throws (src/AssemblerAlign.cpp:419-435).  Either way
    // nothing is stored for the pair; this path reports it as an empty alignment (status EMPTY, not SKIPPED).
    if(offsetMin > offsetMax) return;
    const int32_t bandMin = offsetMin - bandExtend, bandMax = offsetMax + bandExtend;   // :224-225
    if(bandMax - bandMin > maxBand) return;              // :236-241
    const PairDesc pd = pairs[pair];
    DpTask out;
    out.pair = pair;
    out.bandMin = max(bandMin, -int32_t(pd.ny));
    out.bandMax = min(bandMax, int32_t(pd.nx));
    out.label = 0;
    // A band of more than 1024 diagonals (Align.maxBand beyond what the banded DP kernels hold): listed from the back, for the
    // wide DP over the band, as the Align4 cells kernels list such components.
    if(out.bandMax - out.bandMin + 1 > 1024) tasks2[taskCapacity2 - 1u - atomicAdd(taskCount2 + wideCounter, 1u)] = out;
    else tasks2[atomicAdd(taskCount2, 1u)] = out;
}

// Traceback of the wide tasks of Align4 (components of more than 1024 diagonals): one wavefront per task, lane 0 walks the trace
// align3WideDpKernel wrote and stores the aligned pairs from the end of the task's ordinal range downwards, as dpTracebackKernel
// does; results[firstResult + k] gets what tracebackFinish leaves.  A handful of tasks in a run whose Align.maxBand admits them:
// written to be right, not fast.
__global__ void __launch_bounds__(64)
wideTracebackKernel(const PairDesc* __restrict__ pairs, const WideTask* __restrict__ wideTasks, const WideEnd* __restrict__ wideEnds, uint32_t count,
    const uint64_t* __restrict__ trace, const uint64_t* __restrict__ ordBases, uint32_t* __restrict__ ordScratch, DpResult* __restrict__ results, uint32_t firstResult)
{
    const uint32_t k = blockIdx.x;
    if(k >= count || threadIdx.x != 0) return;
    const WideTask task = wideTasks[k];
    const WideEnd e = wideEnds[k];
    const PairDesc pd = pairs[task.pair];
    const uint64_t* __restrict__ tr = trace + task.traceOffset;
    const uint64_t ordBase = ordBases[k];
    uint32_t pos = min(pd.nx, pd.ny);
    int32_t i = e.bestI, j = e.bestJ;
    if(e.score > NEG_SCORE) {
        while(i > 0 && j > 0) {
            const uint32_t b = uint32_t(i - j - task.dMin);
            const uint64_t w = 2ULL * (uint64_t(uint32_t(i + j)) * task.chunks + (b >> 6));
            const uint64_t lo = tr[w], hi = tr[w + 1];
            const uint32_t bit = b & 63u;
            const uint32_t dir = uint32_t((lo >> bit) & 1ULL) | (uint32_t((hi >> bit) & 1ULL) << 1);
            if(dir == 0u) { --pos; *reinterpret_cast<uint2*>(ordScratch + 2 * (ordBase + pos)) = make_uint2(uint32_t(i - 1), uint32_t(j - 1)); }
            i -= (dir != 2u) ? 1 : 0;
            j -= (dir != 3u) ? 1 : 0;
        }
    }
    DpEnd end; end.traceOffset = 0; end.bestI = e.bestI; end.bestJ = e.bestJ; end.score = e.score; end.laneBase = 0; end.bundleIterations = 0; end.pad = 0;
    tracebackFinish(pos, pd, end, ordBase, firstResult + k, results);
}
