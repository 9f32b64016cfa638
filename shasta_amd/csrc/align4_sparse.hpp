// Align4 on MI355X, K10s: the banded overlap alignment of a (candidate, component) task computed from the MATCHES inside its
// band -- a few hundred per task, listed by the cells kernel while it counts them -- instead of from every cell of the band
// (a hundred thousand per task: bandedDpForwardKernel + dpTracebackKernel, align4_dp.hpp).  Included by align4.hip inside its
// anonymous namespace, after align4_dp.hpp.
//
// What it computes (oracle/sparse_chain.hpp states it on the CPU and has the proof): only aligned pairs of EQUAL markers leave
// computeBandedAlignment (/root/reference/src/Align4.cpp:1053-1068).  With scores 6 / -1 / -1 and all end gaps free, a stretch
// between two consecutive matches costs the Chebyshev distance between them, the way in from the free border costs min(x, y), the
// way out min(nx - 1 - x, ny - 1 - y); so over the hits of the band, in the order of one of the two ordinals,
//
//     D(m) = 6 + max( -min(x, y),  max over hits m' before m in BOTH ordinals of  D(m') - max(x - x' - 1, y - y' - 1) )
//     best = max over m of  D(m) - min(nx - 1 - x, ny - 1 - y),          Z = the best score of a path without any match
//
// is the dense DP's optimum, and the dense traceback returns the match set of SOME optimal chain whatever its tie policy.  The
// kernel counts the optimal chains (capped at two) and answers only when there is exactly ONE (or when no chain reaches Z: the
// empty alignment): then every tie policy -- every reading of SeqAn -- gives that very set, and the task is "certified".  A task
// with several optimal chains goes on to sparseAnchorKernel (align4_anchor.hpp: the dense DP only between the matches every
// optimal chain holds), for which this kernel leaves a link word beside every hit; the rest (a hit whose search for predecessors
// goes further back than the ring holds, lists that did not fit, reads beyond the tables here) runs in the dense kernels exactly
// as before.  At 100 k reads 87 % of the tasks (85 % of the DP cells) are certified and all but 0.2 % of the others are the anchor
// kernel's (oracle's census over the same candidates, profiles/r04_sparse_census.txt).
//
//   sparseSortKernel    a wavefront per task: the candidate's hit list filtered by the task's band and ordered by the ordinal in
//                       the TABLED read (fewer than 8192 markers by the cells stage's classes, however long the other read is;
//                       either order serves the recurrence, it is symmetric): a
//                       counting sort on 4-bit counters per marker in LDS (two hits of one marker inside a band are common: the
//                       background of a 15 000-k-mer alphabet; sixteen send the task to the dense DP).
//   sparseChainKernel   a LANE per task (the recurrence is sequential in the hits).  A turn of its loop looks at up to two
//                       predecessors of the lane's hit and, if that settles it, finishes the hit and takes the next: the common
//                       hit is one turn and the lanes stay in step, while a hit that needs a long look back (an off-chain hit: as
//                       many hits back as the band is wide) only holds its own lane.  The last 56 hits of every lane in an LDS
//                       ring {ordinals, D, prefix maximum of D} whose other 8 slots hold the hits the lane needs next; the chosen
//                       predecessor of a hit written back into the task's list, which the lane then walks from the best end to
//                       emit the pairs into the task's range of the ordinal scratch, as dpTracebackKernel does.
//   dpDenseFlagsKernel / dpDenseListKernel   the sorted task list without the certified tasks, and the class counts of what is left.
#pragma once

// Markers of the read the hits are ordered by (4-bit counters: 1.25 bytes of LDS per marker and wavefront): the tabled read -- below
// 8 192 in the cells stage's first four classes, the shorter read of the candidate in the windowed one (round 6).  32 768: the list word
// of a finished hit holds the ordinal in 15 bits, and D of a hit of such a read still fits the wave kernel's 16 bits (6 - min(p, s) > -32 768).
constexpr uint32_t SPARSE_MAX_STREAM = 32768;
constexpr int SPARSE_RING = 64;                              // slots of a lane's ring: the hits it can look back on and the next ones it will need
constexpr int SPARSE_LOOK_BACK = 56;                         // hits a lane can look back
enum SparseState : uint8_t { SPARSE_DENSE = 0, SPARSE_CERTIFIED = 1, SPARSE_SORTED = 2, SPARSE_AMBIGUOUS = 3, SPARSE_COMPLETE = 4, SPARSE_COMPLETE_STREAM = 5 };   // COMPLETE_STREAM: COMPLETE, and the alignment in shasta::compress form at the end of the task's room in the list of sorted hits (the wave kernel's; compressWriteKernel copies it)
//   // CERTIFIED: the pairs are in the ordinal scratch; COMPLETE: and the task's metrics in its result (dpMetricsKernel skips it);   // AMBIGUOUS: several optimal chains, forward pass kept: sparseAnchorKernel's (align4_anchor.hpp)
constexpr int SPARSE_LINK_REACH = 29;                        // how far back (in hits) a hit's link word names its optimal links; bit 30: one goes further

// Where task t's ordered hits go: room for 2 min(nx, ny) + 64 of them (ordOffsets[t] = where the task's min(nx, ny) + 32 pairs of
// the ordinal scratch begin, dpSizeKernel: twice that, in words, is a range of its own for every task).
__host__ __device__ inline uint64_t sparseListBase(const uint64_t* ordOffsets, uint32_t t) { return 2 * ordOffsets[t]; }
__host__ __device__ inline uint32_t sparseListCapacity(uint32_t nx, uint32_t ny) { return 2u * (nx < ny ? nx : ny) + 64u; }
__device__ __forceinline__ uint32_t nibbleSum(uint32_t v)
{
    v = (v & 0x0f0f0f0fu) + ((v >> 4) & 0x0f0f0f0fu);
    return (v * 0x01010101u) >> 24;
}

// The best score of a path without a match: the largest -min(i, j) over the border cells (i = nx or j = ny) inside the band
// (oracle::bestMatchlessScore).
__host__ __device__ inline int32_t sparseMatchlessScore(uint32_t nx, uint32_t ny, int32_t bandMin, int32_t bandMax)
{
    long long best = -(1LL << 40);
    {
        const long long jLo = (long long)nx - bandMax > 0 ? (long long)nx - bandMax : 0, jHi = (long long)nx - bandMin < (long long)ny ? (long long)nx - bandMin : (long long)ny;
        if(jLo <= jHi) { const long long v = -(jLo < (long long)nx ? jLo : (long long)nx); best = v > best ? v : best; }
    }
    {
        const long long iLo = (long long)ny + bandMin > 0 ? (long long)ny + bandMin : 0, iHi = (long long)ny + bandMax < (long long)nx ? (long long)ny + bandMax : (long long)nx;
        if(iLo <= iHi) { const long long v = -(iLo < (long long)ny ? iLo : (long long)ny); best = v > best ? v : best; }
    }
    return int32_t(best > -(1LL << 30) ? best : -(1LL << 30));
}

// align4_chainwave.hpp's capacity classes (hits a wavefront holds in LDS).
constexpr int CHAIN_WAVE_CLASSES = 6;
// (a little under the powers of two: with the 4 bytes per window of 64 hits beside the 10 per hit, 1 024 hits were 10 304 bytes a
// wavefront -- fifteen to a CU's 160 KB instead of sixteen)
// (round 6: two classes between 4 064 and 15 360 hits -- 5 456, the most whose D fits 16 bits: three wavefronts to a CU, and 8 000 with D in
// 32 bits: two -- for the tasks of two long reads, 4 500 hits on average in the ultra-long shape: half of them ran one wavefront to a CU)
constexpr uint32_t CHAIN_WAVE_CAPACITY[CHAIN_WAVE_CLASSES] = {1016u, 2032u, 4064u, 5456u, 8000u, 15360u};
// The classes from this one on are few tasks each (0.1 % of the tasks at 100 k reads): the sort kernel LISTS them (an atomic each on the
// class's counter), and their launches run the list instead of going over all the tasks for them.
constexpr int CHAIN_WAVE_LISTED_FROM = 2;
__host__ __device__ inline int chainWaveClassOf(uint32_t hits)
{
    for(int c = 0; c < CHAIN_WAVE_CLASSES; c++) if(hits <= CHAIN_WAVE_CAPACITY[c]) return c;
    return -1;
}

__device__ __forceinline__ void noteGiveUp(DpControl* control, int why, const PairDesc& pd, const DpTask& task)
{
    atomicAdd(&control->giveUpTasks[why], 1u);
    atomicAdd(&control->giveUpCells[why], (unsigned long long)pd.nx * (unsigned long long)(task.bandMax - task.bandMin + 1));
}

// Two launches by the markers of the read the order is by: up to 4 096 (5 KB of counters a wavefront: 32 wavefronts per CU; 99 % of
// the tasks at 100 k reads) and beyond (10 KB, 16 per CU).  The kernel waits for memory, it does not compute: twice the wavefronts
// in flight is what it needed (one launch sized for 8 192 markers: 14 ms per step alone, 72 ms of launches sharing the device).
// (round 6) ... and two more for the windowed class's candidates: up to 16 384 markers (two wavefronts a workgroup, 20 KB each) and up to
// 32 768 (one, 40 KB); launched only for a batch that has such candidates.
template<int MAX_STREAM, int MIN_STREAM, int WAVES = 4>
__global__ void __launch_bounds__(64 * WAVES)
sparseSortKernel(const PairDesc* __restrict__ pairs, const DpTask* __restrict__ tasks, uint32_t taskCount,
    const uint32_t* __restrict__ hits, const uint64_t* __restrict__ hitBase, const uint32_t* __restrict__ hitMeta,
    const uint64_t* __restrict__ ordOffsets, uint32_t* __restrict__ sorted, uint32_t* __restrict__ inBand, uint8_t* __restrict__ state, DpControl* __restrict__ control, bool noLaneKernel,
    uint32_t* __restrict__ waveLists)
{
    static_assert(MAX_STREAM <= int(SPARSE_MAX_STREAM) && MIN_STREAM < MAX_STREAM && MAX_STREAM % 8 == 0, "classes of the tabled read's markers");
    __shared__ uint32_t counts[WAVES][MAX_STREAM / 8], cursors[WAVES][MAX_STREAM / 8];
    __shared__ uint16_t wordStart[WAVES][MAX_STREAM / 8];
    const int lane = laneId();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * uint32_t(WAVES) + wave;
    if(t >= taskCount) return;                             // (whole wavefronts: no block barrier below)
    const DpTask task = tasks[t];
    const PairDesc pd = pairs[task.pair];
    const uint32_t meta = hitMeta[task.pair];
    const bool swapped = (meta >> 31) != 0;
    const uint32_t count = meta & 0x7fffffffu;
    const uint64_t begin = hitBase[task.pair];
    const uint32_t capacity = uint32_t(hitBase[task.pair + 1] - begin);
    // The order is by the ordinal in the TABLED read (read 1 if the chunk was swapped): it has fewer than 8192 markers whenever a
    // chunk kernel made the list (the LDS table's classes), however long the streamed read is.
    const uint32_t streamCount = swapped ? pd.ny : pd.nx;          // (markers of the read the order is by)
    if(meta == HIT_LIST_NONE || count > capacity || streamCount > SPARSE_MAX_STREAM || streamCount == 0) {
        if(lane == 0 && MIN_STREAM == 0) {                 // (said once: by the first launch)
            state[t] = SPARSE_DENSE;
            noteGiveUp(control, meta == HIT_LIST_NONE ? GIVE_UP_NO_LIST : (count > capacity ? GIVE_UP_LIST_OVERFLOW : GIVE_UP_LONG_STREAM), pd, task);
        }
        return;
    }
    if(streamCount > uint32_t(MAX_STREAM) || streamCount <= uint32_t(MIN_STREAM)) return;      // (the other launch's)
    uint32_t* const myCounts = counts[wave];
    uint32_t* const myCursors = cursors[wave];
    uint16_t* const myStart = wordStart[wave];
    const uint32_t words = (streamCount + 7u) / 8u;
    for(uint32_t w = uint32_t(lane); w < words; w += WAVE) { myCounts[w] = 0; myCursors[w] = 0; }
    waveLdsSync();
    const uint32_t* __restrict__ const list = hits + begin;
    // A list of up to 64 x SORT_HELD hits (nearly all: 850 listed per candidate at 100 k reads) is read ONCE, every lane's share into
    // registers with all the loads in flight together; both passes below then run from the registers.  (Until round 5 each pass read
    // the list 64 hits per step, a step's load issued when the one before had been used: 2 x 13 trips to memory per task, one after
    // the other -- the kernel waited, it did not compute.)  Longer lists: the two passes over memory, as before.
    constexpr int SORT_HELD = 16;
    uint32_t held[SORT_HELD];
    const bool holdsAll = count <= uint32_t(WAVE * SORT_HELD);
    if(holdsAll) {
#pragma unroll
        for(int a = 0; a < SORT_HELD; a++) { const uint32_t i = uint32_t(a * WAVE + lane); held[a] = list[i < count ? i : 0u]; }
    }
    bool crowded = false;
    auto countHit = [&](uint32_t e, bool exists) {
        const int32_t x = int32_t(e >> 16), y = int32_t(e & 0xffffu);
        const bool in = exists && x - y >= task.bandMin && x - y <= task.bandMax;
        const uint32_t p = uint32_t(swapped ? y : x);
        if(in && p < streamCount) {
            const uint32_t shift = 4u * (p & 7u);
            const uint32_t old = atomicAdd(&myCounts[p >> 3], 1u << shift);
            crowded |= ((old >> shift) & 15u) == 15u;
        }
    };
    if(holdsAll) {
#pragma unroll
        for(int a = 0; a < SORT_HELD; a++) if(uint32_t(a * WAVE) < count) countHit(held[a], uint32_t(a * WAVE + lane) < count);
    } else {
        for(uint32_t i0 = 0; i0 < count; i0 += WAVE) { const uint32_t i = i0 + uint32_t(lane); countHit(list[i < count ? i : 0u], i < count); }
    }
    waveLdsSync();
    if(__any(crowded)) { if(lane == 0) { state[t] = SPARSE_DENSE; noteGiveUp(control, GIVE_UP_CROWDED_MARKER, pd, task); } return; }
    // Where every word's markers start: a scan of the words' sums.
    const uint32_t per = (words + WAVE - 1) / WAVE, first = uint32_t(lane) * per;
    uint32_t sum = 0;
    for(uint32_t w = first; w < min(first + per, words); w++) sum += nibbleSum(myCounts[w]);
    uint32_t inclusive = sum;
#pragma unroll
    for(int d = 1; d < WAVE; d <<= 1) { const uint32_t o = uint32_t(__shfl_up(int(inclusive), d, WAVE)); if(lane >= d) inclusive += o; }
    const uint32_t total = uint32_t(__shfl(int(inclusive), WAVE - 1, WAVE));
    if(total > sparseListCapacity(pd.nx, pd.ny) || total > 0xffffu || (noLaneKernel && chainWaveClassOf(total) < 0)) {          // (0xffff: the words' first positions are 16-bit)
        // (noLaneKernel: the wave kernel is on and sparseChainKernel is not launched -- what the wave kernel's largest class does not hold is the dense kernels')
        if(lane == 0) { state[t] = SPARSE_DENSE; noteGiveUp(control, GIVE_UP_SORTED_CAPACITY, pd, task); }
        return;
    }
    uint32_t running = inclusive - sum;
    for(uint32_t w = first; w < min(first + per, words); w++) { myStart[w] = uint16_t(running); running += nibbleSum(myCounts[w]); }
    waveLdsSync();
    uint32_t* __restrict__ const out = sorted + sparseListBase(ordOffsets, t);
    auto placeHit = [&](uint32_t e, bool exists) {
        const int32_t x = int32_t(e >> 16), y = int32_t(e & 0xffffu);
        const bool in = exists && x - y >= task.bandMin && x - y <= task.bandMax;
        const uint32_t p = uint32_t(swapped ? y : x), s = uint32_t(swapped ? x : y);
        if(in && p < streamCount) {
            const uint32_t w = p >> 3, shift = 4u * (p & 7u);
            const uint32_t below = nibbleSum(myCounts[w] & ((1u << shift) - 1u));
            const uint32_t local = (atomicAdd(&myCursors[w], 1u << shift) >> shift) & 15u;
            out[uint32_t(myStart[w]) + below + local] = (p << 16) | s;
        }
    };
    if(holdsAll) {
#pragma unroll
        for(int a = 0; a < SORT_HELD; a++) if(uint32_t(a * WAVE) < count) placeHit(held[a], uint32_t(a * WAVE + lane) < count);
    } else {
        for(uint32_t i0 = 0; i0 < count; i0 += WAVE) { const uint32_t i = i0 + uint32_t(lane); placeHit(list[i < count ? i : 0u], i < count); }
    }
    if(lane == 0) {
        inBand[t] = total; state[t] = SPARSE_SORTED;
        const int cls = chainWaveClassOf(total);
        if(waveLists && cls >= CHAIN_WAVE_LISTED_FROM) waveLists[uint64_t(cls - 1) * taskCount + atomicAdd(&control->retryCount[cls], 1u)] = t;
    }
}

// Ring entry: {p << 16 | s,  (D + SPARSE_D_BIAS) | (prefix maximum - D, 1023 = not held) << 20 | (two or more optimal chains end here) << 30}.
constexpr int32_t SPARSE_D_BIAS = 1 << 17;
constexpr int32_t SPARSE_NEG = -(1 << 29);
// A hit of the task's list once it is finished: p << 17 | (s - p - lo) << 7 | how many hits back its predecessor is (0: the chain starts here).
__global__ void __launch_bounds__(64)
sparseChainKernel(const PairDesc* __restrict__ pairs, const DpTask* __restrict__ tasks, const uint32_t* __restrict__ order, uint32_t taskCount,
    uint32_t* __restrict__ sorted, const uint32_t* __restrict__ inBand, uint8_t* __restrict__ state, const uint32_t* __restrict__ hitMeta,
    const uint64_t* __restrict__ ordOffsets, uint32_t* __restrict__ ordScratch, DpResult* __restrict__ results,
    uint32_t* __restrict__ linkWords, DpEnd* __restrict__ ends, uint32_t* __restrict__ ambiguousList, DpControl* __restrict__ control,
    DeviceOptions opt, unsigned long long* __restrict__ pairBest)
{
    __shared__ uint2 ring[SPARSE_RING * WAVE];             // [slot][lane]
    const int lane = laneId();
    const uint32_t position = blockIdx.x * uint32_t(WAVE) + uint32_t(lane);
    const bool has = position < taskCount;
    const uint32_t t = order[has ? taskCount - 1u - position : 0u];       // the list is ascending in length: the longest tasks first
    const bool mine = has && state[t] == SPARSE_SORTED;
    const DpTask task = tasks[t];
    const PairDesc pd = pairs[task.pair];
    const bool swapped = (hitMeta[task.pair] >> 31) != 0;
    const int32_t n = mine ? int32_t(inBand[t]) : 0;
    const int32_t np = int32_t(swapped ? pd.ny : pd.nx), ns = int32_t(swapped ? pd.nx : pd.ny);     // p: the ordinal in the tabled read, s: in the other
    const int32_t lo = swapped ? task.bandMin : -task.bandMax;            // s - p lies in [lo, lo + band width)
    uint32_t* __restrict__ const list = sorted + sparseListBase(ordOffsets, t);
    // Beside every finished hit, for the tasks that turn out to have several optimal chains (align4_anchor.hpp): WHICH terms attain its
    // maximum -- bit 0 the border, bit d the hit d back (d <= 29; bit 30: a hit further back does) -- and, bit 31, whether a chain that ends
    // with it reaches the best end so far.
    uint32_t* __restrict__ const myLinks = linkWords + sparseListBase(ordOffsets, t);
    // The hits a lane will need next wait in the ring itself: slots k .. k + 7 hold the raw hits k .. k + 7 (the scan looks back
    // SPARSE_LOOK_BACK = 56 hits at most, so the 64 slots hold both).  They arrive in two groups of four, A and B, each with its own
    // place in the loop -- A's every eighth turn, B's four turns later, the same for the whole wavefront: a group is written into
    // the ring when it is the next one and fits (its last hit below k + 8), and asked for again at once, so a group's loads have
    // eight turns to arrive and nothing inside a turn waits for memory (a lane that read its next hit from memory when it finished
    // one -- or kept them in registers that shift along -- waited a memory latency per hit: the compiler can only place the wait
    // where the register is next touched, and that is the very next hit).  A lane that is behind keeps its groups for a later visit.
    //
    // A turn of the loop: up to TWO predecessors looked at (both ring entries read up front), and, if the hit is finished by then,
    // its results written and the next hit taken from the ring.  The common hit -- its predecessor one back, the bound on
    // everything before that below it -- takes ONE turn, and the lanes of a wavefront stay in step; a hit that has to look far
    // back takes a turn for every two entries.  (The first form took a turn for each of: take a hit, look at one entry, finish --
    // three turns for the common hit, and with the lanes in different states every turn paid for all three: 150 vector
    // instructions a turn.)
    uint32_t groupA[4], groupB[4];
    int32_t baseA = 0, countA = 0, baseB = 0, countB = 0;
    bool validA = false, validB = false;
    int32_t fetched = 0, requested = 0;       // hits [0, fetched) are in the ring or consumed; [fetched, requested) are in flight
    {
        const int32_t first = min(n, 8);
        uint32_t head[8];                       // (all eight loads before the first is looked at: one latency, not eight)
#pragma unroll
        for(int a = 0; a < 8; a++) head[a] = list[n > 0 ? min(a, n - 1) : 0];
#pragma unroll
        for(int a = 0; a < 8; a++) if(a < first) ring[a * WAVE + lane].x = head[a];
        fetched = first;
        baseA = fetched; countA = max(0, min(4, n - baseA)); validA = countA > 0;
        baseB = baseA + countA; countB = max(0, min(4, n - baseB)); validB = countB > 0;
        requested = baseB + countB;
#pragma unroll
        for(int a = 0; a < 4; a++) groupA[a] = list[n > 0 ? min(baseA + a, n - 1) : 0];
#pragma unroll
        for(int a = 0; a < 4; a++) groupB[a] = list[n > 0 ? min(baseB + a, n - 1) : 0];
    }
    int32_t k = 0, p = 0, s = 0;
    int32_t value = 0, from = 0, j = 1;
    uint32_t ways = 1;
    uint32_t links = 1;                       // bit 0 the border, bit d the hit d back (d <= SPARSE_LINK_REACH), bit 30: one further back
    int32_t prefixMax = SPARSE_NEG, best = SPARSE_NEG, bestAt = -1;
    uint32_t bestWays = 0;
    bool failed = false, active = n > 0, haveHit = false;
    auto takeHit = [&]() {
        const uint32_t hit = ring[(k & (SPARSE_RING - 1)) * WAVE + lane].x;
        p = int32_t(hit >> 16); s = int32_t(hit & 0xffffu);
        value = -min(p, s); from = 0; j = 1; ways = 1; links = 1;
        haveHit = true;
    };
    auto groupVisit = [&](uint32_t (&group)[4], int32_t& base, int32_t& count, bool& valid) {
        const bool write = valid && base == fetched && fetched + count <= k + 8;
#pragma unroll
        for(int a = 0; a < 4; a++) if(write && a < count) ring[((base + a) & (SPARSE_RING - 1)) * WAVE + lane].x = group[a];
        if(write) { fetched += count; valid = false; }
        if(active && !valid && requested < n) { base = requested; count = min(4, n - requested); requested += count; valid = true; }
        // (Unconditional loads from a clamped position: a group that stays is read again -- the list's entries from `fetched` on are
        // still the raw hits -- and one that is not wanted is not looked at.)
#pragma unroll
        for(int a = 0; a < 4; a++) group[a] = list[n > 0 ? min(base + a, n - 1) : 0];
    };
    auto turnOfTheLoop = [&]() {
        if(active && !haveHit && k < fetched) takeHit();                   // (the first hit, and after a wait for a group)
        if(active && haveHit) {
            // Both entries a turn can look at, read before either is needed.
            const uint2 e0 = ring[((k - j) & (SPARSE_RING - 1)) * WAVE + lane];
            const uint2 e1 = ring[((k - j - 1) & (SPARSE_RING - 1)) * WAVE + lane];
            // (Selects, not branches: the lanes of a wavefront differ in what an entry means to them, and the compiler's
            // structured branches cost more than the arithmetic they skip -- 31 register moves and 14 mask saves per turn.)
            bool finish = false, looking = true;
#pragma unroll
            for(int u = 0; u < 2; u++) {
                const uint2 e = u == 0 ? e0 : e1;
                const int32_t pq = int32_t(e.x >> 16), sq = int32_t(e.x & 0xffffu);
                const int32_t dq = int32_t(e.y & 0xfffffu) - SPARSE_D_BIAS;
                const uint32_t held = (e.y >> 20) & 1023u;
                const int32_t maxUpToQ = held == 1023u ? prefixMax : dq + int32_t(held);
                const bool noMore = j > k;                                  // no hit further back
                const bool tooFar = j > SPARSE_LOOK_BACK;                   // further back than the ring holds: the dense DP takes the task
                // No hit at q or before it can reach `value` (every one of them is at least p - pq - 1 away): `<`, so that ties are seen.
                const bool stop = maxUpToQ - (p - pq - 1) < value;
                const bool fail = looking && !noMore && tooFar;
                const bool consider = looking && !noMore && !tooFar && !stop;
                finish = finish || (looking && (noMore || (!tooFar && stop)));
                failed = failed || fail;
                const bool good = consider && pq < p && sq < s;
                const int32_t candidate = dq - max(p - pq - 1, s - sq - 1);
                const uint32_t waysQ = 1u + ((e.y >> 30) & 1u);
                const uint32_t bit = j <= SPARSE_LINK_REACH ? 1u << j : 0x40000000u;
                const bool better = good && candidate > value, equal = good && candidate == value;
                ways = better ? waysQ : (equal ? min(2u, ways + waysQ) : ways);
                links = better ? bit : (equal ? (links | bit) : links);
                from = better ? j : from;
                value = better ? candidate : value;
                j += consider ? 1 : 0;
                looking = consider;
            }
            if(failed) active = false;
            if(finish && active) {
                const int32_t d = 6 + value;
                const int32_t newMax = max(prefixMax, d);
                const uint32_t held = uint32_t(min(newMax - d, 1023));
                ring[(k & (SPARSE_RING - 1)) * WAVE + lane] = make_uint2((uint32_t(p) << 16) | uint32_t(s),
                    uint32_t(d + SPARSE_D_BIAS) | (held << 20) | ((ways >= 2u ? 1u : 0u) << 30));
                prefixMax = newMax;
                list[k] = (uint32_t(p) << 17) | (uint32_t(s - p - lo) << 7) | uint32_t(from);
                const int32_t end = d - min(np - 1 - p, ns - 1 - s);
                myLinks[k] = links | (end >= best ? 0x80000000u : 0u);
                if(end > best) { best = end; bestAt = k; bestWays = ways; }
                else if(end == best) bestWays = min(2u, bestWays + ways);
                ++k;
                haveHit = false;
                if(k == n) active = false;
                else if(k < fetched) takeHit();
            }
        }
    };
    // Eight turns to a round, the two groups' visits at its two fixed places (each group's registers are touched at ONE place of
    // the program: visited by turn number inside one loop body, the compiler merged the two groups into one set of registers
    // copied back and forth every turn, behind a wait for everything in flight).
    while(__any(active)) {
        groupVisit(groupA, baseA, countA, validA);
        turnOfTheLoop(); turnOfTheLoop(); turnOfTheLoop(); turnOfTheLoop();
        groupVisit(groupB, baseB, countB, validB);
        turnOfTheLoop(); turnOfTheLoop(); turnOfTheLoop(); turnOfTheLoop();
    }
    {
        // What the two kernels read, for the kernel table: matches listed for the tasks' candidates, matches inside the tasks' bands.
        const uint32_t meta = hitMeta[task.pair];
        unsigned long long listed = (has && meta != HIT_LIST_NONE) ? (meta & 0x7fffffffu) : 0u, kept = uint32_t(n);
        for(int d = 32; d >= 1; d >>= 1) { listed += __shfl_down(listed, d, WAVE); kept += __shfl_down(kept, d, WAVE); }
        if(lane == 0) { atomicAdd(&control->hitsListed, listed); atomicAdd(&control->hitsInBand, kept); }
    }
    if(!mine) return;
    const int32_t matchless = sparseMatchlessScore(pd.nx, pd.ny, task.bandMin, task.bandMax);
    const bool empty = n == 0 || best < matchless;
    if(failed || (!empty && best == matchless)) { state[t] = SPARSE_DENSE; noteGiveUp(control, failed ? GIVE_UP_LOOK_BACK : GIVE_UP_TIE_WITH_EMPTY, pd, task); return; }
    if(!empty && bestWays != 1u) {
        // Several optimal chains: sparseAnchorKernel's.  The best score and the first hit that reaches it travel in the task's DpEnd
        // (the dense forward kernel writes it anew if the task ends up there).
        DpEnd e; e.traceOffset = 0; e.bestI = bestAt; e.bestJ = 0; e.score = best; e.laneBase = 0; e.bundleIterations = 0; e.pad = 0;
        ends[t] = e;
        ambiguousList[atomicAdd(&control->ambiguousCount, 1u)] = t;
        state[t] = SPARSE_AMBIGUOUS;
        return;
    }
    // The chain, from its last hit back, into the end of the task's range of the ordinal scratch (as dpTracebackKernel leaves it) --
    // and, on the way, what dpMetricsKernel would read the pairs again for: AlignmentInfo's offsets, skips and drifts
    // (src/Alignment.cpp:67-113) and the size of the alignment in shasta::compress form (a streak of consecutive pairs is a record
    // of its first pair's skips and its length: walking back, a streak is known when the pair before its first one is met).
    const uint64_t ordBase = ordOffsets[t];
    uint32_t pos = min(pd.nx, pd.ny);
    DpResult r;
    r.sumOffset = 0; r.first0 = r.first1 = r.last0 = r.last1 = 0; r.minOffset = 0x7fffffff; r.maxOffset = int32_t(0x80000000);
    r.maxSkip = r.maxDrift = 0; r.passes = 0; r.compressedBytes = 0;
    uint64_t bytes = 0;
    if(!empty) {
        // (Eight entries per round trip to memory: the chain mostly steps back by one, and a walk of dependent single loads --
        // a load's address known only when the one before it has arrived -- would be as long as the chain times the latency.)
        constexpr int WALK = 8;
        int32_t at = bestAt;
        bool more = true, haveLater = false;
        int32_t laterX = 0, laterY = 0;           // the pair met before this one: the next one of the alignment
        uint32_t streak = 1;                      // pairs of the streak that the later pair belongs to, from it on
        for(int32_t rounds = 0; more && rounds <= n; rounds++) {          // (a chain has at most n hits: the walk ends whatever the list holds)
            uint32_t window[WALK];
#pragma unroll
            for(int a = 0; a < WALK; a++) window[a] = list[max(at - a, 0)];
            const int32_t top = at;
#pragma unroll
            for(int a = 0; a < WALK; a++) {
                if(more && top - a == at) {
                    const uint32_t e = window[a];
                    const int32_t hp = int32_t(e >> 17), hs = hp + lo + int32_t((e >> 7) & 1023u);
                    const int32_t x = swapped ? hs : hp, y = swapped ? hp : hs;
                    --pos;
                    *reinterpret_cast<uint2*>(ordScratch + 2 * (ordBase + pos)) = make_uint2(uint32_t(x), uint32_t(y));
                    const int32_t offset = x - y;
                    r.minOffset = min(r.minOffset, offset); r.maxOffset = max(r.maxOffset, offset); r.sumOffset += offset;
                    if(haveLater) {
                        const int32_t skip0 = laterX - x, skip1 = laterY - y;
                        r.maxSkip = max(r.maxSkip, max(uint32_t(skip0), uint32_t(skip1)));
                        const int32_t drift = skip0 - skip1;
                        r.maxDrift = max(r.maxDrift, uint32_t(drift < 0 ? -drift : drift));
                        if(skip0 == 1 && skip1 == 1) ++streak;
                        else { bytes += uint64_t(makeStreakRecord(skip0, skip1, streak).len); streak = 1; }
                    }
                    else { r.last0 = uint32_t(x); r.last1 = uint32_t(y); haveLater = true; }
                    laterX = x; laterY = y;
                    const int32_t back = int32_t(e & 127u);
                    if(back == 0 || back > at || pos == 0) more = false; else at -= back;
                }
            }
        }
        // The first pair: its streak's skips are taken against (0, 0).
        r.first0 = uint32_t(laterX); r.first1 = uint32_t(laterY);
        bytes += uint64_t(makeStreakRecord(laterX, laterY, streak).len);
    }
    r.ordBegin = ordBase + pos;
    r.markerCount = min(pd.nx, pd.ny) - pos;
    r.score = empty ? matchless : best;
    r.compressedBytes = uint32_t(bytes < 0xffffffffULL ? bytes : 0xffffffffULL);
    taskAcceptance(r, pd, task, opt, pairBest);
    results[t] = r;
    state[t] = SPARSE_COMPLETE;
}

// The sorted task list without the certified tasks.  flags[i] = 1 where sorted position i stays (flags[taskCount] = 0, for the scan's total).
__global__ void __launch_bounds__(256)
dpDenseFlagsKernel(const uint32_t* __restrict__ sortedIds, const uint8_t* __restrict__ state, uint32_t taskCount, uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i > taskCount) return;
    const uint8_t st = i < taskCount ? state[sortedIds[i]] : uint8_t(SPARSE_CERTIFIED);
    flags[i] = (st != SPARSE_CERTIFIED && st != SPARSE_COMPLETE && st != SPARSE_COMPLETE_STREAM) ? 1u : 0u;
}

// positions = exclusive scan of the flags.  Class counts and the per-class sums (DP cells, algorithmic bytes) of the tasks that stay:
// what dpSizeKernel computed for all of them (the caller zeroes classCounts and sums[2 ...] first).
__global__ void __launch_bounds__(256)
dpDenseListKernel(const uint32_t* __restrict__ sortedKeys, const uint32_t* __restrict__ sortedIds, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ positions,
    uint32_t taskCount, const DpTask* __restrict__ tasks, const PairDesc* __restrict__ pairs,
    uint32_t* __restrict__ denseKeys, uint32_t* __restrict__ denseIds, uint32_t* __restrict__ classCounts, unsigned long long* __restrict__ sums)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1;
    unsigned long long cells = 0, bytes = 0;
    if(i < taskCount && flags[i]) {
        const uint32_t key = sortedKeys[i], t = sortedIds[i];
        denseKeys[positions[i]] = key; denseIds[positions[i]] = t;
        cls = int(key >> 24);
        const DpTask task = tasks[t];
        const PairDesc pd = pairs[task.pair];
        cells = (unsigned long long)(pd.nx) * (unsigned long long)(task.bandMax - task.bandMin + 1);
        bytes = 4ULL * (uint64_t(pd.nx) + pd.ny);
    }
#pragma unroll
    for(int c = 0; c < DP_CLASSES; c++) {
        const uint64_t votes = __ballot(cls == c);
        if(votes == 0) continue;
        unsigned long long classCells = cls == c ? cells : 0, classBytes = cls == c ? bytes : 0;
        for(int d = 32; d >= 1; d >>= 1) { classCells += __shfl_down(classCells, d, WAVE); classBytes += __shfl_down(classBytes, d, WAVE); }
        if(laneId() == 0) {
            atomicAdd(&classCounts[c], uint32_t(__popcll(votes)));
            atomicAdd(&sums[2 + c], classCells);
            atomicAdd(&sums[2 + DP_CLASSES + c], classBytes);
        }
    }
}
