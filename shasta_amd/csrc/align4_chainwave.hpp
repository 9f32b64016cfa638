// Align4 on MI355X, K10w: the chain recurrence of align4_sparse.hpp with a WAVEFRONT per task and the task's hits in LDS.
// Included by align4.hip inside its anonymous namespace, after align4_sparse.hpp (whose definitions of D, the certificate, the
// list and link words it shares: /root/reference/src/Align4.cpp:993-1088 is what both replace for the tasks they answer).
//
// Why a second form.  sparseChainKernel gives every task a LANE: 64 tasks' lists are read and written four bytes at a time at 64
// unrelated addresses per instruction.  Round 5's first hardware profile (profiles/r05_pmc_100k_reads.json): 6.1 GB written to HBM
// per launch for 1.4 GB of list and link words (partial lines leave the L2 before they fill), 53 ms per step; and a hit whose search
// for predecessors reaches further back than the lane's LDS ring (56 hits) sends the task to the dense kernels -- 0.5 % of the
// tasks, but the long wide ones: 2 % of the DP cells, 75 ms per step of dense kernels launched for a few wavefronts each.
//
// What a wavefront can do that a lane cannot: 99 % of the hits of a task lie on its one optimal chain, each linked to the hit just
// before it (profiles/r04_sparse_census.txt: 712 hits per task, 705 aligned pairs, 1.09 predecessors looked at per hit).  So the
// wavefront takes 64 consecutive hits, ASSUMES that each one's only optimal predecessor is the hit before it, and checks that for
// all 64 at once:
//     D(i) = D(i-1) + 6 - c(i),  c(i) = max(p(i) - p(i-1) - 1, s(i) - s(i-1) - 1)           a prefix sum over the lanes
//     the hit before is dominated (p, s both smaller);
//     no hit further back reaches that:  prefixMax D [.. i-2] - (p(i) - p(i-2) - 1)  <  D(i) - 6      a prefix maximum
//     nor does the border:               -min(p(i), s(i))  <  D(i) - 6
// (strict: a tie would be another optimal predecessor).  The recurrence defines D(i) from the D of earlier hits only, so the lanes
// up to the first one whose check fails hold the true D (their sums and maxima involve lower lanes only) -- they are accepted, with
// `from` = 1 and the certificate's count inherited.  The failing hit is an EXCEPTION: the wavefront scans back over ALL earlier hits,
// 64 per step, every lane one candidate, until the bound says nothing further back can matter -- no ring, no look-back limit -- and
// the next 64 hits start behind it.  About 14 exceptions per task (an off-chain hit and the hit after it, a long insertion).
//
// Then, still in LDS: the chain from the best end back (runs of `from` = 1 are whole bit ranges of a ballot; the exceptions are
// the only sequential steps), the aligned pairs written in order, 64 at a time, and AlignmentInfo's metrics and the size in
// shasta::compress form from ballots over the links between consecutive pairs.  HBM sees the sorted hits once (read) and the pairs
// once (written).  A task with several optimal chains gets its list and link words written out as sparseChainKernel leaves them
// (links of the exceptions recomputed from the final D: 64 candidates per step again) and goes to sparseAnchorKernel.
//
// OWN_SORT (SHASTA_MI355X_CHAIN_WAVE_SORT=1; not the default): the kernel orders the hits itself (sparseSortKernel's counting sort, into
// LDS), so that the ordered hits never go to HBM.  Measured on the MI355X it LOSES: 82 ms per step alone against 49 + 15 for the two
// kernels, 166 ms per step against 143 (profiles/r05_call5*, r05_call3*): the sort's two passes over the candidate's list are chains of
// dependent loads, and a wavefront that holds 10 KB of LDS while it waits for them keeps the next task's wavefront out.
//
// LDS: 8 bytes per hit (ordinals, D in 16 bits, `from` + flags) in the classes whose D fits -- the tabled read has fewer than
// SPARSE_MAX_STREAM = 32 768 markers, so D >= 6 - min(p, s) > -32 768, and D <= 6 per hit of the class's capacity; 10 bytes (D in 32 bits) in
// the last class, and in all of them with SHASTA_MI355X_CHAIN_WAVE_WIDE_D=1 (the form before: 16, 8, 4 wavefronts per CU instead of
// 20, 10, 5).  Four launches by capacity -- 1 016 hits (8 KB a wavefront: 87 % of the
// tasks at 100 k reads), 2 032 (13 %), 4 064 (0.1 %), 15 360 -- the first two of wavefronts that go over the task list in blocks of
// 16 and run the tasks of their class, the last two over the few tasks sparseSortKernel listed for them (a launch that went over all
// tasks for a few hundred of them, at one wavefront per CU, took as long as the first class's); what fits none goes to the dense
// kernels.
// SHASTA_MI355X_CHAIN_WAVE=0: sparseSortKernel + sparseChainKernel, as before this file.
#pragma once

// D in 16 bits where every D of a task of the class fits: above -SPARSE_MAX_STREAM (the border term of a hit, -min(p, s), p an ordinal in
// the tabled read) and at most 6 per hit.
constexpr bool chainWaveNarrowD(uint32_t capacity) { return SPARSE_MAX_STREAM <= 32768u && 6u * capacity + 6u < 32768u; }      // (p < 32 768: 6 - min(p, s) >= -32 761)
constexpr uint32_t chainWaveHitBytes(uint32_t capacity, bool narrow) { return narrow && chainWaveNarrowD(capacity) ? 8u : 10u; }
constexpr size_t chainWaveLdsBytes(uint32_t capacity, bool narrow) { return size_t(capacity) * chainWaveHitBytes(capacity, narrow) + 4u * ((size_t(capacity) + 63u) / 64u); }
// Workgroups of one wavefront, as many as the LDS lets a CU hold.
constexpr uint32_t chainWaveGrid(uint32_t capacity, bool narrow) { return 256u * uint32_t(std::max<size_t>(1u, (160u * 1024u) / chainWaveLdsBytes(capacity, narrow))); }
constexpr uint32_t CHAIN_WAVE_SLACK = 512;         // matches listed for a candidate beyond those inside a task's band, usually fewer than this: the background of the whole matrix, other components
constexpr uint32_t CHAIN_WAVE_BLOCK = 16;          // tasks a wavefront takes from the cursor at a time
constexpr uint32_t CHAIN_OFF_MASK = 0x3fffu, CHAIN_OFF_EXCEPTION = 0x4000u, CHAIN_OFF_WAYS = 0x8000u;
static_assert(CHAIN_WAVE_CAPACITY[CHAIN_WAVE_CLASSES - 1] <= CHAIN_OFF_MASK, "`from` in 14 bits");

// v_mov_b32_dpp with `identity` where the row is masked out or the source lane does not exist.
template<int CTRL, int ROW_MASK> __device__ __forceinline__ int32_t dppOr(int32_t identity, int32_t v)
{
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}
// Inclusive scans over the 64 lanes: row_shr 1, 2, 4, 8 inside the rows of 16, then lane 15 of a row into the next row (rows 1 and
// 3), then lane 31 into rows 2 and 3 -- six DPP operations (the compiler folds the move into the add / max).
__device__ __forceinline__ int32_t waveInclusiveSum(int32_t v)
{
    v += dppOr<0x111, 0xf>(0, v); v += dppOr<0x112, 0xf>(0, v); v += dppOr<0x114, 0xf>(0, v); v += dppOr<0x118, 0xf>(0, v);
    v += dppOr<0x142, 0xa>(0, v); v += dppOr<0x143, 0xc>(0, v);
    return v;
}
constexpr int32_t CHAIN_NEG = -(1 << 29);
__device__ __forceinline__ int32_t waveInclusiveMax(int32_t v)
{
    v = max(v, dppOr<0x111, 0xf>(CHAIN_NEG, v)); v = max(v, dppOr<0x112, 0xf>(CHAIN_NEG, v));
    v = max(v, dppOr<0x114, 0xf>(CHAIN_NEG, v)); v = max(v, dppOr<0x118, 0xf>(CHAIN_NEG, v));
    v = max(v, dppOr<0x142, 0xa>(CHAIN_NEG, v)); v = max(v, dppOr<0x143, 0xc>(CHAIN_NEG, v));
    return v;
}
__device__ __forceinline__ int32_t waveMax(int32_t v) { return int32_t(__builtin_amdgcn_readlane(uint32_t(waveInclusiveMax(v)), WAVE - 1)); }
__device__ __forceinline__ int32_t laneValue(int32_t v, int lane) { return int32_t(__builtin_amdgcn_readlane(uint32_t(v), lane)); }
__device__ __forceinline__ uint64_t bitsUpTo(int b) { return b >= 63 ? ~0ULL : ((2ULL << b) - 1ULL); }       // bits 0 .. b
__device__ __forceinline__ uint64_t bitsAbove(uint64_t m, int lane) { return (m >> lane) >> 1; }              // bits lane + 1 .. 63, moved down to bit 0

// The capacity class of a task, from what is known before its hits are counted: the matches LISTED for its candidate (an upper bound of
// those inside its band) and the markers of the tabled read (the counting sort's 4-bit counters lie where D and `from` will be:
// 1.5 words per hit of capacity, 1 where D is held in 16 bits, for 2.5 words per 8 markers).  -1 and the reason: no class (the dense kernels run the task).
__device__ __forceinline__ int chainWaveClassOfTask(const PairDesc& pd, uint32_t meta, uint64_t room, int& why, bool narrow)
{
    if(meta == HIT_LIST_NONE) { why = GIVE_UP_NO_LIST; return -1; }
    const uint32_t count = meta & 0x7fffffffu;
    if(uint64_t(count) > room) { why = GIVE_UP_LIST_OVERFLOW; return -1; }
    const uint32_t streamCount = (meta >> 31) ? pd.ny : pd.nx;
    if(streamCount > SPARSE_MAX_STREAM || streamCount == 0) { why = GIVE_UP_LONG_STREAM; return -1; }
    const uint32_t counterWords = 5u * ((streamCount + 7u) / 8u);        // in half words: counts, cursors, starts
#pragma unroll
    for(int c = 0; c < CHAIN_WAVE_CLASSES; c++)
        if(count <= CHAIN_WAVE_CAPACITY[c] + CHAIN_WAVE_SLACK && counterWords <= (chainWaveHitBytes(CHAIN_WAVE_CAPACITY[c], narrow) - 4u) / 2u * CHAIN_WAVE_CAPACITY[c]) return c;
    why = GIVE_UP_SORTED_CAPACITY;
    return -1;
}

template<int CAP, bool OWN_SORT, bool NARROW>
__global__ void __launch_bounds__(64)
sparseChainWaveKernel(const PairDesc* __restrict__ pairs, const DpTask* __restrict__ tasks, uint32_t taskCount, int cls,
    DpControl* __restrict__ control, uint32_t* __restrict__ sorted, uint32_t* __restrict__ inBand, uint8_t* __restrict__ state,
    const uint32_t* __restrict__ hits, const uint64_t* __restrict__ hitBase, const uint32_t* __restrict__ hitMeta, const uint64_t* __restrict__ ordOffsets, uint32_t* __restrict__ ordScratch, DpResult* __restrict__ results,
    uint32_t* __restrict__ linkWords, DpEnd* __restrict__ ends, uint32_t* __restrict__ ambiguousList, uint32_t* __restrict__ retryList,
    DeviceOptions opt, unsigned long long* __restrict__ pairBest, uint32_t emitStream)
{
    extern __shared__ uint32_t ldsWords[];
    uint32_t* const H = ldsWords;                                            // p << 16 | s
    // NARROW: the launches of all classes hold D in 16 bits where the class allows it (the classes' capacities for the counting sort
    // of OWN_SORT depend on it, and every launch has to see the same classes).
    constexpr bool NARROW_D = NARROW && chainWaveNarrowD(CAP);
    using StoredD = std::conditional_t<NARROW_D, int16_t, int32_t>;
    constexpr int D_WORDS = NARROW_D ? CAP / 2 : CAP;
    static_assert(size_t(CAP + D_WORDS + CAP / 2 + (CAP + 63) / 64) * 4u == chainWaveLdsBytes(CAP, NARROW), "the arrays below fill exactly what the launch asks for");
    StoredD* const Dv = reinterpret_cast<StoredD*>(ldsWords + CAP);          // D of the finished hits
    uint16_t* const OFF = reinterpret_cast<uint16_t*>(ldsWords + CAP + D_WORDS);   // how many hits back the predecessor is (0: the chain starts here) | flags
    int32_t* const WM = reinterpret_cast<int32_t*>(ldsWords + CAP + D_WORDS + CAP / 2);     // the largest D up to the end of every window of 64 hits
    const int lane = laneId();
    unsigned long long walked = 0, listed = 0;
    // The tasks in blocks of CHAIN_WAVE_BLOCK, taken through a cursor (an atomic per block: a few thousand per launch -- an atomic per
    // TASK on one address, 275 000 per launch, cost the sort kernel 8 ms a step when it appended the tasks to class lists): a lane per
    // task looks at its state and hit count, and the wavefront runs those of its class one after the other.
    uint64_t todoTasks = 0;
    uint32_t blockBase = 0;
    bool scanning = OWN_SORT || cls < CHAIN_WAVE_LISTED_FROM;          // (the few tasks of the larger classes come listed by sparseSortKernel)
    const uint32_t retried = cls > 0 ? control->retryCount[cls] : 0u;       // (listed by the launch of the class below, which has ended)
    uint32_t retrySlot = blockIdx.x;
    for(;;) {
        uint32_t t;
        if(todoTasks) {
            t = blockBase + uint32_t(__ffsll((unsigned long long)todoTasks) - 1);
            todoTasks &= todoTasks - 1;
        } else if(scanning) {
            uint32_t blockIndex = 0;
            if(lane == 0) blockIndex = atomicAdd(&control->waveNext[cls], 1u);
            blockIndex = __builtin_amdgcn_readfirstlane(blockIndex);
            blockBase = blockIndex * CHAIN_WAVE_BLOCK;
            if(blockBase >= taskCount) { scanning = false; continue; }
            const uint32_t candidateTask = blockBase + uint32_t(lane);
            int myClass = -2;                              // -2: not a task of this block; -1: none of the classes holds it (the dense kernels')
            if(!OWN_SORT) {
                // (sparseSortKernel has ordered the hits: the class from their number; what it left to the dense kernels, or what no class holds, is not taken)
                if(uint32_t(lane) < CHAIN_WAVE_BLOCK && candidateTask < taskCount && state[candidateTask] == SPARSE_SORTED) myClass = chainWaveClassOf(inBand[candidateTask]);
            } else
            if(uint32_t(lane) < CHAIN_WAVE_BLOCK && candidateTask < taskCount) {
                const DpTask mine = tasks[candidateTask];
                const PairDesc myPair = pairs[mine.pair];
                int why = -1;
                myClass = chainWaveClassOfTask(myPair, hitMeta[mine.pair], hitBase[mine.pair + 1] - hitBase[mine.pair], why, NARROW);
                if(myClass < 0 && cls == 0) { state[candidateTask] = SPARSE_DENSE; noteGiveUp(control, why, myPair, mine); }      // (said once: by the first class's launch)
            }
            todoTasks = ballot64(myClass == cls);
            continue;
        } else {
            // The tasks whose hits inside the band turned out more than the class below holds.
            if(retrySlot >= retried) break;
            t = retryList[uint64_t(cls - 1) * taskCount + retrySlot];
            retrySlot += gridDim.x;
        }
        const DpTask task = tasks[t];
        const PairDesc pd = pairs[task.pair];
        const bool swapped = (hitMeta[task.pair] >> 31) != 0;
        const int32_t np = int32_t(swapped ? pd.ny : pd.nx), ns = int32_t(swapped ? pd.nx : pd.ny);
        const int32_t lo = swapped ? task.bandMin : -task.bandMax;
        uint32_t* __restrict__ const list = sorted + sparseListBase(ordOffsets, t);
        // ---- the band's hits in the order of the ordinal in the tabled read (what sparseSortKernel does, into LDS instead of HBM):
        // a counting sort on 4-bit counters per marker; the counters lie where D and `from` will be ----
        int32_t n = 0;
        if(!OWN_SORT) {
            n = int32_t(inBand[t]);
            listed += hitMeta[task.pair] & 0x7fffffffu;                     // (what sparseSortKernel read for the task: for the kernel table)
            waveLdsSync();                                                  // (the task before has left the arrays)
            for(int32_t i = lane; i < n; i += WAVE) H[i] = list[i];
            waveLdsSync();
        } else {
            const uint32_t listedHits = hitMeta[task.pair] & 0x7fffffffu;
            const uint32_t* __restrict__ const raw = hits + hitBase[task.pair];
            const uint32_t streamCount = uint32_t(np);
            const uint32_t words = (streamCount + 7u) / 8u;
            uint32_t* const counts = ldsWords + CAP;
            uint32_t* const cursors = counts + words;
            uint16_t* const wordStart = reinterpret_cast<uint16_t*>(cursors + words);
            listed += listedHits;
            waveLdsSync();                                                  // (the task before has left the arrays)
            for(uint32_t w = uint32_t(lane); w < 2u * words; w += WAVE) counts[w] = 0;
            waveLdsSync();
            bool crowded = false;
            for(uint32_t i0 = 0; i0 < listedHits; i0 += WAVE) {
                const uint32_t i = i0 + uint32_t(lane);
                const uint32_t e = raw[i < listedHits ? i : 0u];
                const int32_t x = int32_t(e >> 16), y = int32_t(e & 0xffffu);
                const bool in = i < listedHits && x - y >= task.bandMin && x - y <= task.bandMax;
                const uint32_t p = uint32_t(swapped ? y : x);
                if(in && p < streamCount) {
                    const uint32_t shift = 4u * (p & 7u);
                    const uint32_t old = atomicAdd(&counts[p >> 3], 1u << shift);
                    crowded |= ((old >> shift) & 15u) == 15u;
                }
            }
            waveLdsSync();
            if(__any(crowded)) { if(lane == 0) { state[t] = SPARSE_DENSE; noteGiveUp(control, GIVE_UP_CROWDED_MARKER, pd, task); } continue; }
            const uint32_t per = (words + WAVE - 1) / WAVE, first = uint32_t(lane) * per;
            uint32_t sum = 0;
            for(uint32_t w = first; w < min(first + per, words); w++) sum += nibbleSum(counts[w]);
            const uint32_t inclusive = uint32_t(waveInclusiveSum(int32_t(sum)));
            const uint32_t total = uint32_t(laneValue(int32_t(inclusive), WAVE - 1));
            if(total > uint32_t(CAP) && total <= sparseListCapacity(pd.nx, pd.ny) && cls + 1 < CHAIN_WAVE_CLASSES && total <= CHAIN_WAVE_CAPACITY[CHAIN_WAVE_CLASSES - 1]) {
                // (the class was chosen from the matches listed for the candidate less the usual background: this task has more inside its band)
                if(lane == 0) retryList[uint64_t(cls) * taskCount + atomicAdd(&control->retryCount[cls + 1], 1u)] = t;
                continue;
            }
            if(total > sparseListCapacity(pd.nx, pd.ny) || total > uint32_t(CAP)) { if(lane == 0) { state[t] = SPARSE_DENSE; noteGiveUp(control, GIVE_UP_SORTED_CAPACITY, pd, task); } continue; }
            uint32_t running = inclusive - sum;
            for(uint32_t w = first; w < min(first + per, words); w++) { wordStart[w] = uint16_t(running); running += nibbleSum(counts[w]); }
            waveLdsSync();
            for(uint32_t i0 = 0; i0 < listedHits; i0 += WAVE) {
                const uint32_t i = i0 + uint32_t(lane);
                const uint32_t e = raw[i < listedHits ? i : 0u];
                const int32_t x = int32_t(e >> 16), y = int32_t(e & 0xffffu);
                const bool in = i < listedHits && x - y >= task.bandMin && x - y <= task.bandMax;
                const uint32_t p = uint32_t(swapped ? y : x), sOrdinal = uint32_t(swapped ? x : y);
                if(in && p < streamCount) {
                    const uint32_t w = p >> 3, shift = 4u * (p & 7u);
                    const uint32_t below = nibbleSum(counts[w] & ((1u << shift) - 1u));
                    const uint32_t local = (atomicAdd(&cursors[w], 1u << shift) >> shift) & 15u;
                    H[uint32_t(wordStart[w]) + below + local] = (p << 16) | sOrdinal;
                }
            }
            n = int32_t(total);
            if(lane == 0) inBand[t] = total;
            waveLdsSync();
        }
        walked += uint32_t(n);
        SHASTA_DEVICE_CHECK(n >= 0 && n <= CAP);

        // ---- forward: D, `from`, the count of optimal chains (capped at two) ----
        // Windows of 64 hits at fixed places.  The lanes' steps 6 - c(i) and their prefix sums are made once per window; an exception
        // in the window only shifts the D of the lanes behind it by what its own D turned out to differ from the assumed one, so the
        // window goes on behind it with a new prefix maximum and the same sums (a window restarted behind every exception made 24
        // passes for a task's 11 windows).
#ifdef CHAIN_DEBUG
        int dbgPasses = 0, dbgExceptions = 0, dbgBlocks = 0;
#endif
        int32_t pmAll = CHAIN_NEG, pmBut1 = CHAIN_NEG;                      // the largest D of the hits before the first undecided one, and of those before the last of them
        for(int32_t k = 0; k < n; k += WAVE) {
            const int32_t i = k + lane;
            const bool valid = i < n;
            const uint32_t h = H[valid ? i : n - 1];
            const uint32_t h1 = H[i >= 1 && valid ? i - 1 : 0], h2 = H[i >= 2 && valid ? i - 2 : 0];
            const int32_t p = int32_t(h >> 16), s = int32_t(h & 0xffffu);
            const int32_t a = p - int32_t(h1 >> 16) - 1, b = s - int32_t(h1 & 0xffffu) - 1;
            const bool simple = valid && i >= 1 && a >= 0 && b >= 0;
            const int32_t border = -min(p, s);
            const int32_t reach2 = p - int32_t(h2 >> 16) - 1;                 // what separates the hit from the one two before, at least
            int32_t d = (k > 0 ? Dv[k - 1] : 0) + waveInclusiveSum(simple ? 6 - max(a, b) : 0);
            uint32_t waysBefore = k > 0 ? (uint32_t(OFF[k - 1]) >> 15) : 0u;      // (of the last decided hit: what a run of accepted hits inherits)
            int first = 0;                                                    // the first undecided lane
            for(;;) {
#ifdef CHAIN_DEBUG
                ++dbgPasses;
#endif
                const int32_t pmIncl = max(pmAll, waveInclusiveMax(lane >= first ? d : CHAIN_NEG));
                int32_t pm2 = dppOr<0x138, 0xf>(CHAIN_NEG, dppOr<0x138, 0xf>(CHAIN_NEG, pmIncl));      // the largest D up to the hit two before (wave_shr:1 twice)
                pm2 = lane == first ? pmBut1 : (lane == first + 1 ? pmAll : pm2);
                const int32_t value = d - 6;                                  // D(i - 1) - c(i)
                const bool ok = lane < first || (simple && (i < 2 || pm2 - reach2 < value) && border < value);
                const uint64_t okMask = ballot64(ok);
                const int accepted = okMask == ~0ULL ? WAVE : (__ffsll((unsigned long long)~okMask) - 1);      // lanes [first, accepted) hold their true D
                if(accepted > first) {
                    if(lane >= first && lane < accepted) { SHASTA_DEVICE_CHECK(int32_t(StoredD(d)) == d); Dv[i] = StoredD(d); OFF[i] = uint16_t(1u | (waysBefore ? CHAIN_OFF_WAYS : 0u)); }
                    const int32_t newAll = laneValue(pmIncl, accepted - 1);
                    pmBut1 = accepted - 2 >= first ? laneValue(pmIncl, accepted - 2) : pmAll;
                    pmAll = newAll;
                }
                const int32_t e = k + accepted;
                if(accepted >= WAVE || e >= n) break;
                waveLdsSync();                                                // (the exception's scan reads what the accepted lanes wrote)
                // The exception: hit e against every hit before it, 64 per step from the nearest back.
#ifdef CHAIN_DEBUG
                ++dbgExceptions;
#endif
                const uint32_t he = H[e];
                const int32_t pe = int32_t(he >> 16), se = int32_t(he & 0xffffu);
                int32_t bestValue = -min(pe, se);
                uint32_t ways = 1;
                int32_t from = 0;
                // Exceptions come in pairs: an off-chain hit, and the hit behind it, whose predecessor in the list it is -- and which it
                // does not dominate, or dominates without being its optimal predecessor.  So the hit behind an exception is always
                // decided with it: both take their candidates from the same steps of the scan -- one pass over the earlier hits
                // instead of two, and no checked pass in between (23.2 -> 19 passes and 14.4 -> 9 scan steps per task of 704 hits;
                // pairing only where the next lane was known to fail, by `simple`, caught 3 of a task's 7 pairs).
                const bool pairOf = accepted + 1 < WAVE && e + 1 < n;
                const uint32_t he2 = H[pairOf ? e + 1 : e];
                const int32_t pe2 = int32_t(he2 >> 16), se2 = int32_t(he2 & 0xffffu);
                int32_t bestValue2 = -min(pe2, se2);
                uint32_t ways2 = 1;
                int32_t from2 = 0;
                bool done1 = false, done2 = !pairOf;
                for(int32_t top = e - 1; top >= 0; top -= WAVE) {
                    // Nothing at `top` or before it can reach bestValue: every one of them is at least pe - p(top) - 1 away, and none has a
                    // larger D than the largest up to the end of top's window (of the windows before this one: WM; of this one: so far).
                    const int32_t bound = (top >> 6) < (k >> 6) ? WM[top >> 6] : pmAll;
                    const int32_t pTop = int32_t(H[top] >> 16);
                    done1 = done1 || bound - (pe - pTop - 1) < bestValue;
                    done2 = done2 || bound - (pe2 - pTop - 1) < bestValue2;
                    if(done1 && done2) break;
#ifdef CHAIN_DEBUG
                    ++dbgBlocks;
#endif
                    const int32_t q = top - lane;
                    const uint32_t hq = H[q >= 0 ? q : 0];
                    const int32_t pq = int32_t(hq >> 16), sq = int32_t(hq & 0xffffu);
                    const int32_t dq = Dv[q >= 0 ? q : 0];
                    const bool twoQ = (uint32_t(OFF[q >= 0 ? q : 0]) & CHAIN_OFF_WAYS) != 0;
                    if(!done1) {
                        const bool good = q >= 0 && pq < pe && sq < se;
                        const int32_t candidate = good ? dq - max(pe - pq - 1, se - sq - 1) : CHAIN_NEG;
                        const int32_t blockBest = waveMax(candidate);
                        if(blockBest > bestValue) { bestValue = blockBest; ways = 0; from = -1; }
                        if(blockBest == bestValue) {
                            const uint64_t at = ballot64(candidate == bestValue), atTwo = ballot64(candidate == bestValue && twoQ);
                            ways = min(2u, ways + uint32_t(__popcll(at)) + uint32_t(__popcll(atTwo)));
                            if(from < 0) from = e - top + (__ffsll((unsigned long long)at) - 1);       // the nearest hit that attains it
                        }
                    }
                    if(!done2) {
                        const bool good = q >= 0 && pq < pe2 && sq < se2;
                        const int32_t candidate = good ? dq - max(pe2 - pq - 1, se2 - sq - 1) : CHAIN_NEG;
                        const int32_t blockBest = waveMax(candidate);
                        if(blockBest > bestValue2) { bestValue2 = blockBest; ways2 = 0; from2 = -1; }
                        if(blockBest == bestValue2) {
                            const uint64_t at = ballot64(candidate == bestValue2), atTwo = ballot64(candidate == bestValue2 && twoQ);
                            ways2 = min(2u, ways2 + uint32_t(__popcll(at)) + uint32_t(__popcll(atTwo)));
                            if(from2 < 0) from2 = e + 1 - top + (__ffsll((unsigned long long)at) - 1);
                        }
                    }
                }
                const int32_t de = 6 + bestValue;
                SHASTA_DEVICE_CHECK(int32_t(StoredD(de)) == de);
                if(lane == 0) { Dv[e] = StoredD(de); OFF[e] = uint16_t(uint32_t(from) | CHAIN_OFF_EXCEPTION | (ways >= 2u ? CHAIN_OFF_WAYS : 0u)); }
                pmBut1 = pmAll; pmAll = max(pmAll, de);
                waysBefore = ways >= 2u ? 1u : 0u;
                int last = accepted;                                          // the last lane decided here
                int32_t dLast = de;
                if(pairOf) {
#ifdef CHAIN_DEBUG
                    ++dbgExceptions;
#endif
                    // The second hit's one candidate the scan did not hold: the first hit itself, the nearest of all.
                    if(pe < pe2 && se < se2) {
                        const int32_t candidate = de - max(pe2 - pe - 1, se2 - se - 1);
                        const uint32_t waysE = ways >= 2u ? 2u : 1u;
                        if(candidate > bestValue2) { bestValue2 = candidate; ways2 = waysE; from2 = 1; }
                        else if(candidate == bestValue2) { ways2 = min(2u, ways2 + waysE); if(from2 != 0) from2 = 1; }       // (from2 == 0: the border attains it)
                    }
                    const int32_t de2 = 6 + bestValue2;
                    SHASTA_DEVICE_CHECK(int32_t(StoredD(de2)) == de2);
                    if(lane == 0) { Dv[e + 1] = StoredD(de2); OFF[e + 1] = uint16_t(uint32_t(from2) | CHAIN_OFF_EXCEPTION | (ways2 >= 2u ? CHAIN_OFF_WAYS : 0u)); }
                    pmBut1 = pmAll; pmAll = max(pmAll, de2);
                    waysBefore = ways2 >= 2u ? 1u : 0u;
                    last = accepted + 1; dLast = de2;
                }
                // The lanes behind: their sums were taken from the D assumed for the last hit decided here.
                d += dLast - laneValue(d, last);
                first = last + 1;
                if(first >= WAVE || k + first >= n) break;
            }
            if(lane == 0) WM[k >> 6] = pmAll;                                 // the largest D up to the end of this window
            waveLdsSync();
        }
        // The best end: D - what is left to the border, its first hit, the count of optimal chains that end there (capped at two).
        int32_t best = CHAIN_NEG, bestAt = -1;
        uint32_t bestWays = 0;
        for(int32_t k = 0; k < n; k += WAVE) {
            const int32_t i = k + lane;
            const bool valid = i < n;
            const uint32_t h = H[valid ? i : n - 1];
            const int32_t end = valid ? Dv[i] - min(np - 1 - int32_t(h >> 16), ns - 1 - int32_t(h & 0xffffu)) : CHAIN_NEG;
            const int32_t windowBest = waveMax(end);
            if(windowBest >= best) {
                const uint64_t at = ballot64(end == windowBest), atTwo = ballot64(end == windowBest && (uint32_t(OFF[valid ? i : 0]) & CHAIN_OFF_WAYS) != 0);
                const uint32_t ways = min(2u, uint32_t(__popcll(at)) + uint32_t(__popcll(atTwo)));
                if(windowBest > best) { best = windowBest; bestAt = k + (__ffsll((unsigned long long)at) - 1); bestWays = ways; }
                else bestWays = min(2u, bestWays + ways);
            }
        }
#ifdef CHAIN_DEBUG
        if(lane == 0) std::fprintf(stderr, "chainwave: task %u n %d passes %d exceptions %d blocks %d\n", t, n, dbgPasses, dbgExceptions, dbgBlocks);
#endif
        // ---- what the task is ----
        const int32_t matchless = sparseMatchlessScore(pd.nx, pd.ny, task.bandMin, task.bandMax);
        const bool empty = n == 0 || best < matchless;
        if(!empty && best == matchless) {
            if(lane == 0) { state[t] = SPARSE_DENSE; noteGiveUp(control, GIVE_UP_TIE_WITH_EMPTY, pd, task); }
            continue;
        }
        if(!empty && bestWays != 1u) {
            // Several optimal chains: the list and link words as sparseChainKernel leaves them, for sparseAnchorKernel.  A link word:
            // bit 0 the border attains the hit's maximum, bit d the hit d back (d <= 29; bit 30: one further back); bit 31: a chain
            // that ends with the hit reaches the best end so far.
            uint32_t* __restrict__ const myLinks = linkWords + sparseListBase(ordOffsets, t);
            int32_t pmRunning = CHAIN_NEG, endRunning = CHAIN_NEG;
            for(int32_t base = 0; base < n; base += WAVE) {
                const int32_t i = base + lane;
                const bool valid = i < n;
                const uint32_t h = H[valid ? i : n - 1];
                const int32_t p = int32_t(h >> 16), s = int32_t(h & 0xffffu);
                const int32_t d = valid ? Dv[i] : CHAIN_NEG;
                const uint32_t off = valid ? uint32_t(OFF[i]) : 1u;
                const int32_t pmIncl = max(pmRunning, waveInclusiveMax(d));
                const int32_t end = valid ? d - min(np - 1 - p, ns - 1 - s) : CHAIN_NEG;
                const int32_t endIncl = max(endRunning, waveInclusiveMax(end));
                int32_t endBefore = dppOr<0x138, 0xf>(CHAIN_NEG, endIncl);
                if(lane == 0) endBefore = endRunning;
                uint32_t links = 2u;                                          // an accepted hit: the hit before it, nothing else
                uint64_t todo = ballot64(valid && (off & CHAIN_OFF_EXCEPTION) != 0);
                while(todo) {
                    const int j = __ffsll((unsigned long long)todo) - 1;
                    todo &= todo - 1;
                    const int32_t e = base + j;
                    const uint32_t he = H[e];
                    const int32_t pe = int32_t(he >> 16), se = int32_t(he & 0xffffu);
                    const int32_t target = Dv[e] - 6;
                    const int32_t pmBefore = j == 0 ? pmRunning : laneValue(pmIncl, j - 1);
                    uint32_t word = -min(pe, se) == target ? 1u : 0u;
                    for(int32_t top = e - 1; top >= 0; top -= WAVE) {
                        if(pmBefore - (pe - int32_t(H[top] >> 16) - 1) < target) break;
                        const int32_t q = top - lane;
                        const uint32_t hq = H[q >= 0 ? q : 0];
                        const int32_t pq = int32_t(hq >> 16), sq = int32_t(hq & 0xffffu);
                        const bool attains = q >= 0 && pq < pe && sq < se && Dv[q] - max(pe - pq - 1, se - sq - 1) == target;
                        const uint64_t at = ballot64(attains);
                        if(at) {
                            const int32_t nearest = e - top;                  // lane 0's distance
                            // lane l is the hit nearest + l back
                            if(nearest <= SPARSE_LINK_REACH) word |= uint32_t((at << nearest) & 0x3ffffffeULL);
                            const int reach = SPARSE_LINK_REACH - nearest;    // lanes beyond it are further back than the word names
                            if(reach < 0 || (reach < 63 && (at >> (reach + 1)) != 0)) word |= 0x40000000u;
                        }
                    }
                    if(lane == j) links = word;
                }
                if(valid) {
                    myLinks[i] = links | (end >= endBefore ? 0x80000000u : 0u);
                    list[i] = (uint32_t(p) << 17) | (uint32_t(s - p - lo) << 7) | min(off & CHAIN_OFF_MASK, 127u);
                }
                pmRunning = laneValue(pmIncl, WAVE - 1);
                endRunning = laneValue(endIncl, WAVE - 1);
            }
            if(lane == 0) {
                DpEnd e; e.traceOffset = 0; e.bestI = bestAt; e.bestJ = 0; e.score = best; e.laneBase = 0; e.bundleIterations = 0; e.pad = 0;
                ends[t] = e;
                ambiguousList[atomicAdd(&control->ambiguousCount, 1u)] = t;
                state[t] = SPARSE_AMBIGUOUS;
            }
            continue;
        }

        // ---- the chain, its pairs, AlignmentInfo's metrics (src/Alignment.cpp:67-113), its size in shasta::compress form ----
        const uint64_t ordBase = ordOffsets[t];
        uint32_t pos = min(pd.nx, pd.ny);
        int32_t minOffset = 0x7fffffff, maxOffset = int32_t(0x80000000);
        long long sumOffset = 0;
        uint32_t maxSkip = 0, maxDrift = 0;
        unsigned long long bytes = 0;
        // emitStream: the streaks are written as they are met, from the last one back, so that the alignment in shasta::compress form ends
        // where the task's room in the list of sorted hits ends (the hits are in LDS by now; a streak takes at most 8 bytes -- ordinals
        // have 16 bits -- and the room is 8 bytes per marker of the shorter read: it always fits).  compressWriteKernel copies it.
        uint8_t* const streamEnd = reinterpret_cast<uint8_t*>(list + sparseListCapacity(pd.nx, pd.ny));
        uint32_t tail = 0;                    // bytes written so far
        int32_t laterX = 0, laterY = 0, lastX = 0, lastY = 0;                 // the lowest pair met so far (the successor of the next one met); the last pair of the alignment
        bool haveLater = false;
        uint32_t carryLength = 0;             // pairs from the lowest one met up to and including the first one that is followed by a new streak (or up to the end)
        int32_t cur = empty ? -1 : bestAt;
        while(cur >= 0) {
            const int32_t base = (cur >> 6) << 6;
            const int32_t i = base + lane;
            const bool valid = i < n;
            const uint32_t off = valid ? (uint32_t(OFF[i]) & CHAIN_OFF_MASK) : 1u;
            const uint32_t h = H[valid ? i : n - 1];
            const uint64_t notOne = ballot64(valid && off != 1u);
            uint64_t on = 0;
            while(cur >= base) {
                const int top = cur - base;
                const uint64_t below = notOne & bitsUpTo(top);
                if(below == 0) { on |= bitsUpTo(top); cur = base - 1; break; }
                const int j = 63 - __clzll((unsigned long long)below);
                on |= bitsUpTo(top) & ~(bitsUpTo(j) >> 1);
                const int32_t back = laneValue(int32_t(off), j);
                cur = back == 0 ? -1 : base + j - back;
            }
            const bool isOn = ((on >> lane) & 1ULL) != 0;
            const uint32_t count = uint32_t(__popcll(on));
            const uint32_t rank = uint32_t(__popcll(on & laneMaskLt()));
            const int32_t hp = int32_t(h >> 16), hs = int32_t(h & 0xffffu);
            const int32_t x = swapped ? hs : hp, y = swapped ? hp : hs;
            pos -= count;
            // (emitStream bit 1: nobody reads the aligned pairs of this task -- the caller did not ask for the ordinals and the streaks below are what
            // compressWriteKernel copies: 8 bytes per pair, 1 GB per launch at 100 k reads, not written)
            if(isOn && !(emitStream & 2u)) *reinterpret_cast<uint2*>(ordScratch + 2 * (ordBase + pos + rank)) = make_uint2(uint32_t(x), uint32_t(y));
            // The pair after this one: the next lane up that is on the chain, or the lowest pair of the part met before.
            const uint64_t above = bitsAbove(on, lane);
            const int nextLane = above ? lane + __ffsll((unsigned long long)above) : lane;
            int32_t nextX = __shfl(x, nextLane, WAVE), nextY = __shfl(y, nextLane, WAVE);
            const bool hasNext = above != 0 || haveLater;
            if(above == 0) { nextX = laterX; nextY = laterY; }
            bool newStreak = false;
            int32_t skip0 = 0, skip1 = 0;
            if(isOn) {
                const int32_t offset = x - y;
                minOffset = min(minOffset, offset); maxOffset = max(maxOffset, offset); sumOffset += offset;
                if(hasNext) {
                    skip0 = nextX - x; skip1 = nextY - y;
                    maxSkip = max(maxSkip, max(uint32_t(skip0), uint32_t(skip1)));
                    const int32_t drift = skip0 - skip1;
                    maxDrift = max(maxDrift, uint32_t(drift < 0 ? -drift : drift));
                    newStreak = !(skip0 == 1 && skip1 == 1);
                }
            }
            const uint64_t starts = ballot64(newStreak);      // lanes whose successor begins a streak
            StreakRecord record;
            uint32_t recordBytes = 0;
            if(newStreak) {
                // The streak that begins with the successor: up to and including the next pair that is followed by a new one.
                const uint64_t startsAbove = bitsAbove(starts, lane);
                uint32_t length;
                if(startsAbove) {
                    const int until = lane + __ffsll((unsigned long long)startsAbove);
                    length = uint32_t(__popcll(on & bitsUpTo(until) & ~bitsUpTo(lane)));
                } else length = uint32_t(__popcll(above)) + carryLength;
                record = makeStreakRecord(skip0, skip1, length);
                recordBytes = uint32_t(record.len);
                bytes += uint64_t(recordBytes);
            }
            if((emitStream & 1u) && starts) {
                // (in the order of the lanes: a lane's streak lies behind those of the lanes below it)
                const uint32_t inclusive = uint32_t(waveInclusiveSum(int32_t(recordBytes)));
                const uint32_t total = uint32_t(laneValue(int32_t(inclusive), WAVE - 1));
                SHASTA_DEVICE_CHECK(tail + total <= 4u * sparseListCapacity(pd.nx, pd.ny));
                if(newStreak) writeStreakRecord(record, streamEnd - tail - (total - inclusive) - recordBytes);
                tail += total;
            }
            if(starts) carryLength = uint32_t(__popcll(on & bitsUpTo(__ffsll((unsigned long long)starts) - 1)));
            else carryLength += count;
            if(on) {
                if(!haveLater) { const int topLane = 63 - __clzll((unsigned long long)on); lastX = laneValue(x, topLane); lastY = laneValue(y, topLane); }
                const int lowLane = __ffsll((unsigned long long)on) - 1;
                laterX = laneValue(x, lowLane); laterY = laneValue(y, lowLane);
                haveLater = true;
            }
        }
        // Sums over the lanes.
#pragma unroll
        for(int dlt = 32; dlt >= 1; dlt >>= 1) {
            minOffset = min(minOffset, __shfl_xor(minOffset, dlt, WAVE)); maxOffset = max(maxOffset, __shfl_xor(maxOffset, dlt, WAVE));
            sumOffset += __shfl_xor(sumOffset, dlt, WAVE);
            maxSkip = max(maxSkip, uint32_t(__shfl_xor(int(maxSkip), dlt, WAVE))); maxDrift = max(maxDrift, uint32_t(__shfl_xor(int(maxDrift), dlt, WAVE)));
            bytes += __shfl_xor(bytes, dlt, WAVE);
        }
        if(lane == 0) {
            DpResult r;
            r.sumOffset = 0; r.first0 = r.first1 = r.last0 = r.last1 = 0; r.minOffset = 0x7fffffff; r.maxOffset = int32_t(0x80000000);
            r.maxSkip = r.maxDrift = 0; r.passes = 0; r.compressedBytes = 0;
            if(haveLater) {
                // (the first pair: its streak's skips are taken against (0, 0))
                const StreakRecord record = makeStreakRecord(laterX, laterY, carryLength);
                SHASTA_DEVICE_CHECK(!(emitStream & 1u) || (bytes == tail && tail + uint32_t(record.len) <= 4u * sparseListCapacity(pd.nx, pd.ny)));
                if(emitStream & 1u) writeStreakRecord(record, streamEnd - tail - uint32_t(record.len));
                bytes += uint64_t(record.len);
                r.sumOffset = sumOffset; r.minOffset = minOffset; r.maxOffset = maxOffset; r.maxSkip = maxSkip; r.maxDrift = maxDrift;
                r.first0 = uint32_t(laterX); r.first1 = uint32_t(laterY); r.last0 = uint32_t(lastX); r.last1 = uint32_t(lastY);
            }
            r.ordBegin = ordBase + pos;
            r.markerCount = min(pd.nx, pd.ny) - pos;
            r.score = empty ? matchless : best;
            r.compressedBytes = uint32_t(bytes < 0xffffffffULL ? bytes : 0xffffffffULL);
            taskAcceptance(r, pd, task, opt, pairBest);
            results[t] = r;
            state[t] = (emitStream & 1u) ? SPARSE_COMPLETE_STREAM : SPARSE_COMPLETE;
        }
    }
    if(lane == 0 && walked) { atomicAdd(&control->hitsInBand, walked); atomicAdd(&control->hitsListed, listed); }
}
