// Align4 on MI355X, K10: the banded overlap DP of every (candidate, component) task -- task geometry and sort keys,
// bundles, the forward kernel, the traceback (/root/reference/src/Align4.cpp:993-1088; the SeqAn call it wraps is
// restated, tie policy as in oracle/banded_dp.hpp).  Included by align4.hip inside its anonymous namespace; align
// method 3 (align3.hpp) runs the same kernels.
#pragma once

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
// Band classes.  A task of class c occupies G lanes of a wavefront, a lane owns C adjacent diagonals: G C = 32, 48, 64, 80,
// 128, 256, 512, 1024 diagonals.  Band widths are multiples of deltaY (10 by default): at 100 k reads the DP cells fall on
// widths 30 (8 %), 40 (19 %), 50 (23 %), 60 (19 %), 70 (13 %), 80 (8 %), 90-120 (9 %) (profiles/r02_call21_band_widths.log);
// with power-of-two classes only, 28 % of the lanes computed nothing (40 of 64, 70 of 128 ...), hence the classes of 12 and 20
// lanes (five and three tasks per wavefront, 60 of its 64 lanes).
// Four diagonals per lane wherever the band allows it: a lane's cells of one anti-diagonal are independent of each other, and
// the neighbour exchange (a DPP move per anti-diagonal, plus a select at the group edges unless the group is a DPP row or
// the whole wavefront) is shared by twice the cells -- 9.3 instead of 11.4 VALU instructions per cell.
constexpr int DP_CLASSES = 8;
__host__ __device__ inline int dpClassOfWidth(int32_t w)
{
    return w <= 32 ? 0 : (w <= 48 ? 1 : (w <= 64 ? 2 : (w <= 80 ? 3 : (w <= 128 ? 4 : (w <= 256 ? 5 : (w <= 512 ? 6 : 7))))));
}
__host__ __device__ inline int dpLanes(int cls)                                                              // G = 16,12,16,20,32,64,64,64
{
    return cls == 0 ? 16 : (cls == 1 ? 12 : (cls == 2 ? 16 : (cls == 3 ? 20 : (cls == 4 ? 32 : 64))));
}
__host__ __device__ inline int dpDiagonalsLog2(int cls) { return cls == 0 ? 1 : (cls <= 5 ? 2 : cls - 3); }   // C = 2,4,4,4,4,4,8,16
__host__ __device__ inline int dpDiagonals(int cls) { return 1 << dpDiagonalsLog2(cls); }

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

// The DP tasks' sort key: (class, iterations) -- 3 bits of class over 24 bits of iterations (the host refuses pairs with
// nx + ny >= 2^25 - 4, so the count fits).  One ascending run per class: the forward kernel and the traceback take a class's
// list from its end, longest first.  (Round 2 shipped (class, band width a multiple of C or not, iterations) for a steady loop
// that switched lanes without diagonals off instead of masking their cells: 6.8 instead of 7.8 VALU instructions per cell by
// the ISA, and 83 -> 95 ms per step for the forward launches once it was timed -- two ascending runs per class start the
// second run's longest bundles in the middle of a launch; with one run, or with the two interleaved in steps of 64
// iterations, the loop itself gained nothing measurable: 83.1 / 83.8 / 84.0 ms, profiles/r03_forward_dp_ab.log.  Both are gone.)
__host__ __device__ inline uint32_t dpSortKey(int cls, uint32_t iters) { return (uint32_t(cls) << 24) | (iters < 0xffffffu ? iters : 0xffffffu); }
__host__ __device__ inline uint32_t dpSortKeyIterations(uint32_t key) { return key & 0xffffffu; }
constexpr int DP_SORT_KEY_BITS = 27;

// What the forward kernel leaves for the traceback of a task.
struct DpEnd { uint64_t traceOffset; int32_t bestI, bestJ, score; uint32_t laneBase, bundleIterations, pad; };   // bundleIterations: of the longest task of the bundle

// The tasks of a batch are put in the order (class, length) by ONE counting pass, not by a radix sort: nothing needs the order to
// be exact or stable -- a bundle wants tasks of about the same length (its trace is laid out for its longest), a launch wants its
// longest bundles first -- and a radix sort of 3e5 keys is twenty launches of a few microseconds each that wait, one after the
// other, behind the other workers' kernels (round 3: 0.5 ms alone, 7 ms in flight per batch).  A task's bin: its class and its
// iteration count to one part in 64 (exact below 256); the order inside a bin is whatever the atomics make it.
constexpr int DP_BINS_PER_CLASS = 2048, DP_BINS = DP_CLASSES * DP_BINS_PER_CLASS;
__host__ __device__ inline uint32_t dpBinOfKey(uint32_t key)
{
    const uint32_t cls = key >> 24, iters = dpSortKeyIterations(key);
    uint32_t q = iters;
    if(iters >= 256u) {
        const uint32_t e = 31u - uint32_t(__builtin_clz(iters));            // 8 .. 23
        q = 256u + (e - 8u) * 64u + ((iters >> (e - 6u)) & 63u);
    }
    return cls * uint32_t(DP_BINS_PER_CLASS) + q;
}
// Why a task of the sparse path ends in the dense kernels (counted on the device, reported in the kernel table as rows without time).
enum DpGiveUp : int { GIVE_UP_NO_LIST = 0, GIVE_UP_LIST_OVERFLOW, GIVE_UP_LONG_STREAM, GIVE_UP_CROWDED_MARKER, GIVE_UP_SORTED_CAPACITY,
    GIVE_UP_LOOK_BACK, GIVE_UP_TIE_WITH_EMPTY, GIVE_UP_FAR_LINK, GIVE_UP_NO_ANCHOR, GIVE_UP_WINDOWS, GIVE_UP_RECTANGLE, GIVE_UP_ANCHORS_OFF,
    GIVE_UP_RECTANGLE_PAIRS, GIVE_UP_RECTANGLE_WALK, GIVE_UP_PAIR_TOTAL };
constexpr int DP_GIVE_UP_REASONS = 16;
const char* const DP_GIVE_UP_NAMES[DP_GIVE_UP_REASONS] = {
    "dense DP because: the candidate's matches were not listed (HBM-scratch cells kernel)", "dense DP because: the candidate's match list overflowed",
    "dense DP because: the tabled read has more than 32768 markers", "dense DP because: 16 matches of one marker inside the band",
    "dense DP because: more matches inside the band than the task's list holds", "dense DP because: a match's predecessors lie further back than the chain kernel looks",
    "dense DP because: the best chain ties with the empty alignment", "dense DP because: an optimal link of a live match beyond its link word",
    "dense DP because: no match lies on every optimal chain", "dense DP because: more than 128 windows between anchors",
    "dense DP because: a rectangle between anchors beyond the anchor kernel's limits", "dense DP because: several optimal chains and the anchor kernel is switched off",
    "dense DP because: more aligned pairs inside the rectangles than the anchor kernel holds", "dense DP because: a rectangle's walk did not reach its fixed corner",
    "dense DP because: anchors and rectangle pairs exceed the shorter read", "dense DP because: (unused)"};
// Everything the DP's preparation counts, in one block of device memory (one memset before, one copy to the host after).
struct DpControl {
    unsigned long long sums[2 + 2 * DP_CLASSES];     // [0] DP cells of all tasks, [1] unused, [2+c] cells of class c, [2+DP_CLASSES+c] bytes of class c (of the tasks the dense kernels run)
    unsigned long long ordCursor, traceCursor;       // aligned pairs reserved for the tasks; trace words laid out for the bundles
    uint32_t classCounts[DP_CLASSES];                // tasks per class (of the tasks the dense kernels run)
    uint32_t ambiguousCount, pad;                    // tasks sparseChainKernel left to sparseAnchorKernel (align4_sparse.hpp, align4_anchor.hpp)
    unsigned long long hitsListed, hitsInBand;       // matches the sparse kernels read from the candidates' lists / kept inside the tasks' bands
    unsigned long long ambiguousHits;                // matches of the tasks sparseAnchorKernel walked
    unsigned long long giveUpCells[DP_GIVE_UP_REASONS];   // DP cells of the tasks the sparse kernels left to the dense ones, by reason (DpGiveUp)
    uint32_t giveUpTasks[DP_GIVE_UP_REASONS];             // ... and how many tasks
    uint32_t bins[DP_BINS];                          // tasks per bin, then (dpBinScanKernel) the bin's first position
    uint32_t cursors[DP_BINS];
    uint32_t anchorBigCount, anchorPad;              // align4_anchor.hpp: tasks with a rectangle beyond the first launch's LDS, listed for the second
    uint32_t retryCount[8];                          // align4_chainwave.hpp: tasks a class's launch found too large for it, listed for the next class's
    uint32_t waveNext[8];                            // align4_chainwave.hpp: the cursor through which the wavefronts of a capacity class take their blocks of tasks
};
constexpr size_t DP_CONTROL_HEAD_BYTES = offsetof(DpControl, bins);

// Per task: key (class, iterations), its bin's count, its range of the ordinal scratch (min(nx, ny) + 32 pairs, handed out by a cursor:
// a wavefront takes its tasks' total in one atomic -- the ranges are in no order), statistics.
__global__ void __launch_bounds__(256)
dpSizeKernel(const DpTask* __restrict__ tasks, const PairDesc* __restrict__ pairs, uint32_t taskCount,
    uint32_t* __restrict__ keys, uint64_t* __restrict__ ordOffsets, DpControl* __restrict__ control, int countClasses)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long cells = 0, bytes = 0;
    uint32_t room = 0;
    int cls = -1;
    if(t < taskCount) {
        const DpTask task = tasks[t];
        const PairDesc pd = pairs[task.pair];
        const DpGeometry g = dpGeometry(task.bandMin, task.bandMax, pd.nx, pd.ny);
        cls = g.cls;
        const uint32_t key = dpSortKey(g.cls, g.iters);
        keys[t] = key;
        atomicAdd(&control->bins[dpBinOfKey(key)], 1u);
        room = min(pd.nx, pd.ny) + 32u;          // (the 32 beyond the alignment's longest: the sparse path's list of the task's hits takes twice this room, align4_sparse.hpp)
        cells = (unsigned long long)(pd.nx) * (unsigned long long)(task.bandMax - task.bandMin + 1);
        bytes = 4ULL * (uint64_t(pd.nx) + pd.ny);
    }
    {
        uint32_t inclusive = room;
#pragma unroll
        for(int d = 1; d < WAVE; d <<= 1) { const uint32_t o = uint32_t(__shfl_up(int(inclusive), d, WAVE)); if(laneId() >= d) inclusive += o; }
        unsigned long long base = 0;
        if(laneId() == WAVE - 1 && inclusive) base = atomicAdd(&control->ordCursor, (unsigned long long)inclusive);
        base = __shfl(base, WAVE - 1, WAVE);
        if(t < taskCount) ordOffsets[t] = base + inclusive - room;
    }
    // Per class: one atomic per wavefront and class present in it (tasks of a wave mostly share a
    // class after the cells kernels); one atomic per task serialises the whole launch on six addresses.
    // (countClasses = 0: the dense list's kernel counts the classes, of the tasks the sparse path leaves.)
    if(countClasses) {
#pragma unroll
        for(int c = 0; c < DP_CLASSES; c++) {
            const uint64_t votes = __ballot(cls == c);
            if(votes == 0) continue;
            unsigned long long classCells = cls == c ? cells : 0, classBytes = cls == c ? bytes : 0;
            for(int d = 32; d >= 1; d >>= 1) { classCells += __shfl_down(classCells, d, WAVE); classBytes += __shfl_down(classBytes, d, WAVE); }
            if(laneId() == 0) {
                atomicAdd(&control->classCounts[c], uint32_t(__popcll(votes)));
                atomicAdd(&control->sums[2 + c], classCells);
                atomicAdd(&control->sums[2 + DP_CLASSES + c], classBytes);
            }
        }
    }
    for(int d = 32; d >= 1; d >>= 1) cells += __shfl_down(cells, d, WAVE);
    if(laneId() == 0 && cells) atomicAdd(&control->sums[0], cells);
}

// bins[] -> first positions (one workgroup: 16384 bins are 16 per thread).
__global__ void __launch_bounds__(1024)
dpBinScanKernel(DpControl* __restrict__ control)
{
    __shared__ uint32_t waveSums[16];
    constexpr int PER = DP_BINS / 1024;
    const int lane = laneId(), wave = int(threadIdx.x) >> 6;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for(int k = 0; k < PER; k++) { v[k] = control->bins[threadIdx.x * PER + k]; sum += v[k]; }
    uint32_t inclusive = sum;
#pragma unroll
    for(int d = 1; d < WAVE; d <<= 1) { const uint32_t o = uint32_t(__shfl_up(int(inclusive), d, WAVE)); if(lane >= d) inclusive += o; }
    if(lane == WAVE - 1) waveSums[wave] = inclusive;
    __syncthreads();
    uint32_t running = inclusive - sum;
    for(int w = 0; w < wave; w++) running += waveSums[w];
#pragma unroll
    for(int k = 0; k < PER; k++) { control->bins[threadIdx.x * PER + k] = running; running += v[k]; }
}

// Every task to its place: its bin's first position + the next free one of the bin.
__global__ void __launch_bounds__(256)
dpScatterKernel(const uint32_t* __restrict__ keys, uint32_t taskCount, DpControl* __restrict__ control, uint32_t* __restrict__ sortedKeys, uint32_t* __restrict__ sortedIds)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= taskCount) return;
    const uint32_t key = keys[t], bin = dpBinOfKey(key);
    const uint32_t at = control->bins[bin] + atomicAdd(&control->cursors[bin], 1u);
    sortedKeys[at] = key; sortedIds[at] = t;
}

// Trace words of each bundle (64/G consecutive tasks of the sorted list of one class).
struct DpClassLayout { uint32_t taskStart[DP_CLASSES + 1]; uint32_t bundleStart[DP_CLASSES + 1]; };

__host__ __device__ inline DpClassLayout dpClassLayout(const uint32_t* classCounts)
{
    DpClassLayout layout;
    layout.taskStart[0] = 0; layout.bundleStart[0] = 0;
    for(int c = 0; c < DP_CLASSES; c++) {
        const uint32_t T = 64u / uint32_t(dpLanes(c));
        layout.taskStart[c + 1] = layout.taskStart[c] + classCounts[c];
        layout.bundleStart[c + 1] = layout.bundleStart[c] + (classCounts[c] + T - 1) / T;
    }
    return layout;
}

// One thread per possible bundle (`capacity` of them: the host does not know the class counts yet); the class layout from the
// counts on the device.  A bundle's trace: room for its longest task (any of them may be: the order inside a bin is not by
// length), rounded to 256 bytes so that the traceback's chunks are whole cache lines; where it lies: handed out by a cursor, a
// wavefront's bundles in one atomic.
__global__ void __launch_bounds__(256)
dpBundleKernel(const uint32_t* __restrict__ sortedKeys, DpControl* __restrict__ control, uint32_t capacity, uint64_t* __restrict__ bundleOffsets)
{
    const uint32_t bundle = blockIdx.x * blockDim.x + threadIdx.x;
    const DpClassLayout layout = dpClassLayout(control->classCounts);
    const uint32_t total = layout.bundleStart[DP_CLASSES];
    unsigned long long words = 0;
    if(bundle < total && bundle <= capacity) {
        int cls = 0;
        while(bundle >= layout.bundleStart[cls + 1]) ++cls;
        const uint32_t T = 64u / uint32_t(dpLanes(cls));
        const uint32_t first = layout.taskStart[cls] + (bundle - layout.bundleStart[cls]) * T;
        const uint32_t end = min(first + T, layout.taskStart[cls + 1]);
        uint32_t iterations = 0;
        for(uint32_t k = first; k < end; k++) iterations = max(iterations, dpSortKeyIterations(sortedKeys[k]));
        words = (uint64_t(iterations) * uint64_t(2 * dpDiagonals(cls)) + 31) & ~31ULL;
    }
    unsigned long long inclusive = words;
#pragma unroll
    for(int d = 1; d < WAVE; d <<= 1) { const unsigned long long o = __shfl_up(inclusive, d, WAVE); if(laneId() >= d) inclusive += o; }
    unsigned long long base = 0;
    if(laneId() == WAVE - 1 && inclusive) base = atomicAdd(&control->traceCursor, inclusive);
    base = __shfl(base, WAVE - 1, WAVE);
    if(bundle < total && bundle <= capacity) bundleOffsets[bundle] = base + inclusive - words;
}

// ---- tie policy --------------------------------------------------------------------------------
// What SeqAn decides inside seqan::globalAlignment and no test of the reference pins (SeqAn is an un-vendored dependency that
// is absent here, oracle/banded_dp.hpp): which predecessor a cell keeps when two of them tie, and which border cell ends the
// alignment when several tie on the maximum.  ONE place: the kernels take the policy as a template parameter, numbered as
// oracle::tiePolicyByIndex -- TIE = 2 * order + (last maximum wins), order over the six priority orders of (diagonal,
// vertical, horizontal): 0 DVH, 1 DHV, 2 VDH, 3 VHD, 4 HDV, 5 HVD.  TIE = 0 is the reading of SeqAn 2.4.0 that the oracle
// restates (dp_formula_linear.h: the diagonal candidate first, replaced by the vertical and then by the horizontal one only on
// a strictly larger score; dp_scout.h: a later border cell replaces the best only if strictly greater, cells visited
// column by column).  If a SeqAn run ever disagrees (oracle/seqan_pin), SHASTA_DP_TIE_POLICY is the line to change; two
// alternatives are compiled beside it and parity-tested against the oracle under the same policy (tests/test_gpu_tie_policy.py)
// so that the switch is known to work: SHASTA_MI355X_DP_TIE_POLICY=<n> selects one at run time, any other value is an error.
#ifndef SHASTA_DP_TIE_POLICY
#define SHASTA_DP_TIE_POLICY 0
#endif
// Compiled beside it, parity-tested against the oracle under the same policy and selected at run time by
// SHASTA_MI355X_DP_TIE_POLICY=<n>: the nearest other readings -- 2 = diagonal >= horizontal >= vertical, first maximum (the
// other way to read `_maxScore(left, right)`), 3 = the same with the last maximum (exercises the other end-cell scan) -- and
// the shipped reading 0 when the build's policy is another (oracle/seqan_pin says which one a SeqAn run follows).
constexpr int DP_TIE_POLICY = SHASTA_DP_TIE_POLICY;
constexpr int dpTieAlternative(int k)
{
    const int candidates[3] = {0, 2, 3};
    for(int c = 0; c < 3; c++) { if(candidates[c] == DP_TIE_POLICY) continue; if(k-- == 0) return candidates[c]; }
    return -1;
}
constexpr int DP_TIE_ALTERNATIVE_A = dpTieAlternative(0), DP_TIE_ALTERNATIVE_B = dpTieAlternative(1);
static_assert(DP_TIE_POLICY >= 0 && DP_TIE_POLICY < 12 && DP_TIE_ALTERNATIVE_A >= 0 && DP_TIE_ALTERNATIVE_B >= 0, "tie policies");
inline bool dpTieCompiled(int tie) { return tie == DP_TIE_POLICY || tie == DP_TIE_ALTERNATIVE_A || tie == DP_TIE_ALTERNATIVE_B; }
template<int V> struct DpTieTag { static constexpr int value = V; };
// f(DpTieTag<TIE>) for the compiled policy `tie`.
template<class F> inline void withDpTie(int tie, F&& f)
{
    if(tie == DP_TIE_POLICY) f(DpTieTag<DP_TIE_POLICY>{});
    else if(tie == DP_TIE_ALTERNATIVE_A) f(DpTieTag<DP_TIE_ALTERNATIVE_A>{});
    else if(tie == DP_TIE_ALTERNATIVE_B) f(DpTieTag<DP_TIE_ALTERNATIVE_B>{});
    else throw std::runtime_error("DP tie policy " + std::to_string(tie) + " is not compiled into this build.");
}
template<int TIE> struct DpTie {
    enum Move { DIAGONAL = 0, VERTICAL = 1, HORIZONTAL = 2 };
    static constexpr int order = TIE / 2;
    static constexpr bool lastMaximum = (TIE & 1) != 0;
    static constexpr int first = order <= 1 ? DIAGONAL : (order <= 3 ? VERTICAL : HORIZONTAL);
    static constexpr int second = (order == 2 || order == 4) ? DIAGONAL : ((order == 0 || order == 5) ? VERTICAL : HORIZONTAL);
    static constexpr int third = 3 - first - second;
    // kept[m]: the cells (a bit per lane) whose move is m, from equal[m] = "candidate m equals the maximum of the three"
    // (only equal[first] and equal[second] are read): the first of the order that reaches the maximum.
    __host__ __device__ static void keep(const uint64_t (&equal)[3], uint64_t (&kept)[3])
    {
        kept[first] = equal[first]; kept[second] = equal[second] & ~equal[first]; kept[third] = ~(equal[first] | equal[second]);
    }
    // Border cell (i, j) against the best so far at equal score: the first maximum in the scan order (columns i ascending,
    // rows j ascending inside a column) is the lexicographically smallest (i, j), the last one the largest.
    __host__ __device__ static bool endCellWins(int32_t i, int32_t j, int32_t bestI, int32_t bestJ)
    {
        return lastMaximum ? (i > bestI || (i == bestI && j > bestJ)) : (i < bestI || (i == bestI && j < bestJ));
    }
    static constexpr int32_t noEndCell = lastMaximum ? -1 : 0x7fffffff;
};

// ---- forward kernel ---------------------------------------------------------------------------
// (The round-1 kernel it replaced -- 75 VALU instructions per iteration, 830 GCUPS on the dominant class against 1360 --
// is gone; what follows is what reading that kernel's ISA led to.)
//  * three phases, general / steady / general.  In the steady phase (all but about a band width
//    of iterations at either end) every cell of the wavefront that exists is inside the matrix and
//    past the first cell of its diagonal, and every kmer-id load is in range: no validity tests,
//    no index clamps.  It runs in blocks of DP_BLOCK iterations, fully unrolled: a lane's kmer ids
//    of a block are DP_BLOCK + C/2 consecutive elements per read, fetched as one 16-byte load per
//    read and block, one block ahead -- no sliding register windows, no per-iteration address;
//  * scores are kept biased by -NEG_SCORE, so "outside the band" is 0 and the neighbour exchange
//    is one DPP shift with zero fill (row_shr/shl for 16-lane groups, wave_shr/shl otherwise)
//    instead of ds_bpermute + select -- same decisions: max, compare and adding a constant commute
//    with the bias, and nothing overflows (|score| < 2^27, bias 2^29, i + j < 2^25);
//  * and by the gap penalties of the cell's anti-diagonal (stored = score + bias - gap (i + j)): the three candidates of a
//    cell then are stored(diagonal) + 8 or + 1, stored(vertical), stored(horizontal) -- the addition of the gap penalty
//    to every cell is gone (9 -> 8 VALU instructions per cell);
//  * trace planes: one ballot per comparison (the mask v_cmp wrote anyway), combined on the
//    scalar unit; the ballot of a combined predicate is compiled to v_cndmask + v_cmp;
//  * the trace record goes from the scalar registers the ballots are in straight to memory: scalar stores
//    (s_store_dwordx4, RW / 2 per iteration), no vector instruction at all.  (Round 1 staged the 256-byte line in LDS:
//    12 % of the wave cycles were LDS issue stalls; the first version of round 2 assembled it in one vector register
//    with v_writelane_b32, 2 RW VALU instructions per iteration of a kernel bound by VALU issue: 8 of 31 for C = 2.
//    scripts/microbench/trace_store.hip measured both ways of writing the same records: identical bytes, 0.84 -> 0.62 ms.)
//  * trip counts are made scalar (readfirstlane), so loop control runs on the scalar unit.
constexpr int DP_BLOCK = 4;
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// The value of the lane below / above in a G-lane group; 0 at the group's edge.
template<int G> __device__ __forceinline__ int32_t fromLaneBelow(int32_t v, int l)
{
    constexpr int ctrl = (G == 16) ? 0x111 : 0x138;             // row_shr:1 : wave_shr:1; bound_ctrl = zero fill
    int32_t r = __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true);
    if constexpr (G != 16 && G != 64) r = (l == 0) ? 0 : r;     // a group that is neither a DPP row nor the wavefront: its own edge
    return r;
}
template<int G> __device__ __forceinline__ int32_t fromLaneAbove(int32_t v, int l)
{
    constexpr int ctrl = (G == 16) ? 0x101 : 0x130;             // row_shl:1 : wave_shl:1
    int32_t r = __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true);
    if constexpr (G != 16 && G != 64) r = (l == G - 1) ? 0 : r;
    return r;
}
struct __attribute__((packed, aligned(4))) KmerQuad { uint32_t v[4]; };     // four consecutive kmer ids, 4-byte aligned
// Which loop a cell is computed in: the general one or the steady one.
template<bool STEADY> struct DpPhase { static constexpr bool value = STEADY; };

// Room for six wavefronts per SIMD (80 vector registers) in the classes of up to four diagonals per lane.
#ifdef __HIPCC__
#define SHASTA_DP_FORWARD_OCCUPANCY(C) __attribute__((amdgpu_waves_per_eu((C) <= 4 ? 6 : 1)))
#else
#define SHASTA_DP_FORWARD_OCCUPANCY(C)      // (the wave64 emulator of tests/emu compiles this file as plain C++)
#endif
// The scores of the recurrence.  Align4 hard-wires 6 / -1 / -1 (src/Align4.hpp:159-161: compile-time constants here, SCORES =
// false); align method 3 takes them from its options (src/AssemblerAlign3.cpp:22-33, 120, 257): any other triple runs the SCORES
// = true instances, the same code with the three values in scalar registers (the host checks that nothing can overflow).
struct DpScores { int32_t match, mismatch, gap; };
template<int G, int C, int TIE, bool SCORES = false>
__global__ void __launch_bounds__(256) SHASTA_DP_FORWARD_OCCUPANCY(C)
bandedDpForwardKernel(
    const uint32_t* __restrict__ kmerIds, const PairDesc* __restrict__ pairs, const DpTask* __restrict__ tasks,
    const uint32_t* __restrict__ sortedIds, uint32_t taskCount,
    const uint64_t* __restrict__ bundleOffsets, uint32_t bundleCount,
    uint64_t* __restrict__ trace, DpEnd* __restrict__ ends, DpScores scores)
{
    const int32_t matchScore = SCORES ? scores.match : MATCH_SCORE, mismatchScore = SCORES ? scores.mismatch : MISMATCH_SCORE, gapScore = SCORES ? scores.gap : GAP_SCORE;
    constexpr int T = WAVE / G, HC = C / 2, RW = 2 * C, U = DP_BLOCK;
    constexpr int AL = 2 * U;                             // steady iterations come in groups of AL: two blocks, one on each of the two register sets of kmer ids
    constexpr int32_t BIAS = -NEG_SCORE, NO_DIAGONAL = 0x40000000;
    static_assert(C >= 2 && C <= 16 && U >= 3 && AL % U == 0, "block geometry");
    const int lane = laneId();
    const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if(slot >= bundleCount) return;                       // whole wave leaves: all 64 lanes are active below, no block barriers
    const uint32_t bundle = bundleCount - 1 - slot;       // the list is sorted by ascending length: the longest bundles start first
    const int g = lane / G, l = lane % G;
    const uint32_t pos = bundle * T + uint32_t(g);
    const bool hasTask = g < T && pos < taskCount;        // (64 - T G lanes of a wavefront are idle when G does not divide 64)
    const uint32_t t = sortedIds[hasTask ? pos : bundle * T];
    const DpTask task = tasks[t];
    const PairDesc pd = pairs[task.pair];
    const uint32_t* __restrict__ p0 = kmerIds + pd.begin0;
    const uint32_t* __restrict__ p1 = kmerIds + pd.begin1;
    const int32_t nx = int32_t(pd.nx), ny = int32_t(pd.ny);
    const int32_t bandMin = task.bandMin, width = task.bandMax - task.bandMin + 1;
    const DpGeometry geo = dpGeometry(task.bandMin, task.bandMax, pd.nx, pd.ny);
    uint32_t itersLane = geo.iters;
    itersLane = hasTask ? itersLane : 0u;
#pragma unroll
    for(int d = 1; d < WAVE; d <<= 1) itersLane = max(itersLane, uint32_t(__shfl_xor(int(itersLane), d, WAVE)));
    const uint32_t iters = __builtin_amdgcn_readfirstlane(itersLane);
    uint64_t* __restrict__ tr = uniformPointer(trace + bundleOffsets[bundle]);

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
    int32_t H[C];                                         // biased: score + BIAS - GAP_SCORE (i + j) of the diagonal's latest cell; 0 = no cell
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
        // Stored values are score + BIAS - GAP_SCORE (i + j): a gap move (one anti-diagonal on, one gap penalty) leaves the
        // stored value as it is, a diagonal move (two anti-diagonals on) adds the match or mismatch score minus two gap
        // penalties -- one addition per cell instead of two, same comparisons (all three candidates carry the same offset).
        const int32_t dg = hd + (eq ? matchScore - 2 * gapScore : mismatchScore - 2 * gapScore);
        // hv: from (i, j-1), diagonal b+1; hh: from (i-1, j), diagonal b-1.  The move kept is the first in the tie policy's order
        // of (diagonal, vertical, horizontal) that equals their maximum (for the restated SeqAn policy: vertical only if
        // hv > dg, horizontal only if hh > max(dg, hv)): one v_max3 and two equality tests instead of two maxima and two comparisons.
        int32_t v = max(max(dg, hv), hh);
        const int32_t candidates[3] = {dg, hv, hh};
        const bool equalsFirst = v == candidates[DpTie<TIE>::first], equalsSecond = v == candidates[DpTie<TIE>::second];
        if constexpr (STEADY) {
            SHASTA_DEVICE_CHECK(!exists[c] || (sc > lo[c] && uint32_t(sc - lo[c]) <= span[c]));
            H[c] = exists[c] ? v : 0;
        } else {
            const bool valid = uint32_t(sc - lo[c]) <= span[c];
            v = (sc == lo[c]) ? BIAS - gapScore * sc : v;           // i == 0 or j == 0: free leading gaps (score 0)
            H[c] = valid ? v : H[c];
        }
        uint64_t equal[3] = {0, 0, 0}, kept[3];
        equal[DpTie<TIE>::first] = ballot64(equalsFirst); equal[DpTie<TIE>::second] = ballot64(equalsSecond);
        DpTie<TIE>::keep(equal, kept);
        loPlane = kept[DpTie<TIE>::HORIZONTAL] | (kept[DpTie<TIE>::DIAGONAL] & ~ballot64(eq));     // codes: 0 diagonal+equal, 1 diagonal+different, 2 vertical, 3 horizontal
        hiPlane = ~kept[DpTie<TIE>::DIAGONAL];
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
    // The record of iteration `it`: RW ballot words, uniform across the wavefront, in scalar registers; they go to
    // tr[it RW ...] as they are.
    auto putRecord = [&](uint32_t it, const uint64_t (&words)[RW]) {
        SHASTA_DEVICE_CHECK((uint64_t(it) + 1) * RW <= ((uint64_t(iters) * RW + 31) & ~31ULL));       // inside the bundle's trace (dpBundleKernel)
#pragma unroll
        for(int k = 0; k < RW; k += 2) scalarStore128(tr + uint64_t(it) * RW + k, words[k], words[k + 1]);
    };
    // The same for iteration `index` (a constant once the steady phase is unrolled) of the group whose records start at `base`.
    auto putRecordOfGroup = [&](uint64_t* base, int index, const uint64_t (&words)[RW]) {
#pragma unroll
        for(int k = 0; k < RW; k += 2) scalarStore128At(base, (index * RW + k) * 8, words[k], words[k + 1]);
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
            antiDiagonals(DpPhase<false>{}, geo.s0 + 2 * int32_t(it), [&](int k) { return aw[k]; }, [&](int h) { return bw[h]; }, words);
            putRecord(it, words);
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
        auto steady = [&](auto phaseTag) {
            // Block registers.  Logically a[x] = A[ib + l HC - 1 + x], x = 0..U+HC-1, and e[x] = B[ib - bandMin - l HC - HC + x],
            // x = 0..U+HC-2; iteration u of the block takes aw(k) = a[u + k], bw(h) = e[u + HC - 1 - h].  The last U elements of
            // each are the 16-byte load of the block (quadA / quadB [blk & 1]: two register sets, the load of the next block goes
            // into the other one, a block ahead), the first HC (HC - 1) are the end of the previous block's, kept in tailA /
            // tailB: HC + HC - 1 register moves per block where copying the prefetched quads into one array took 2 U more
            // (8 of 75 VALU instructions of a block for C = 2).
            const int32_t ib = ib0 + int32_t(steadyBegin);
            uint32_t tailA[HC], tailB[HC > 1 ? HC - 1 : 1];
            KmerQuad quadA[2], quadB[2];
#pragma unroll
            for(int x = 0; x < U + HC; x++) { const uint32_t v = loadA(ib + l * HC - 1 + x); if(x < HC) tailA[x] = v; else quadA[0].v[x - HC] = v; }
#pragma unroll
            for(int x = 0; x < U + HC - 1; x++) { const uint32_t v = loadB(ib - bandMin - l * HC - HC + x); if(x < HC - 1) tailB[x] = v; else quadB[0].v[x - (HC - 1)] = v; }
            const uint32_t* __restrict__ pa = p0 + (int64_t(iaBlock) + int64_t(steadyBegin));
            const uint32_t* __restrict__ pb = p1 + (int64_t(jbBlock) + int64_t(steadyBegin));
            uint64_t* groupRecords = tr + uint64_t(steadyBegin) * RW;
            SHASTA_DEVICE_CHECK((uint64_t(steadyBegin) + uint64_t(groups) * AL) * RW <= ((uint64_t(iters) * RW + 31) & ~31ULL));
            for(uint32_t grp = 0; grp < groups; grp++, groupRecords += AL * RW) {
#pragma unroll
                for(int blk = 0; blk < AL / U; blk++) {
                    const int cur = blk & 1, nxt = cur ^ 1;
                    SHASTA_DEVICE_CHECK(pa >= p0 && pa + U <= p0 + nx && pb >= p1 && pb + U <= p1 + ny);
                    quadA[nxt] = *reinterpret_cast<const KmerQuad*>(pa);
                    quadB[nxt] = *reinterpret_cast<const KmerQuad*>(pb);
                    pa += U; pb += U;
                    auto a = [&](int x) { return x < HC ? tailA[x < HC ? x : 0] : quadA[cur].v[x < HC ? 0 : x - HC]; };
                    auto e = [&](int x) { return x < HC - 1 ? tailB[x < HC - 1 ? x : 0] : quadB[cur].v[x < HC - 1 ? 0 : x - (HC - 1)]; };
#pragma unroll
                    for(int u = 0; u < U; u++) {
                        uint64_t words[RW];
                        antiDiagonals(phaseTag, geo.s0 + 2 * int32_t(steadyBegin + grp * AL + blk * U + u), [&](int k) { return a(u + k); }, [&](int h) { return e(u + HC - 1 - h); }, words);
                        putRecordOfGroup(groupRecords, blk * U + u, words);
                    }
#pragma unroll
                    for(int x = 0; x < HC; x++) tailA[x] = a(x + U);
#pragma unroll
                    for(int x = 0; x < HC - 1; x++) tailB[x] = e(x + U);
                }
            }
        };
        steady(DpPhase<true>{});
        general(steadyBegin + groups * AL, iters);
    }
    scalarStoreFlush();                                   // the scalar data cache is write-back

    // End cell: maximum over the border cells = final value of every diagonal; ties by the policy (the reading: the smallest (i, j)).
    int32_t bestScore = NEG_SCORE, bestI = DpTie<TIE>::noEndCell, bestJ = DpTie<TIE>::noEndCell;
#pragma unroll
    for(int c = 0; c < C; c++) {
        const int32_t d = bandMin + l * C + c;
        const int32_t i = (d >= nx - ny) ? nx : ny + d, j = i - d;
        const int32_t v = exists[c] ? H[c] - BIAS + gapScore * (i + j) : NEG_SCORE;
        if(v > bestScore || (v == bestScore && v > NEG_SCORE && DpTie<TIE>::endCellWins(i, j, bestI, bestJ))) { bestScore = v; bestI = i; bestJ = j; }
    }
    if constexpr ((G & (G - 1)) == 0) {
#pragma unroll
        for(int d = G / 2; d >= 1; d >>= 1) {
            const int32_t os = __shfl_xor(bestScore, d, G);
            const int32_t oi = __shfl_xor(bestI, d, G);
            const int32_t oj = __shfl_xor(bestJ, d, G);
            if(os > bestScore || (os == bestScore && DpTie<TIE>::endCellWins(oi, oj, bestI, bestJ))) { bestScore = os; bestI = oi; bestJ = oj; }
        }
    } else {
        // A group of 12 or 20 lanes: every lane looks at the other lanes of its group in turn (the order is total, so every
        // lane ends with the same cell).
        const int32_t ownScore = bestScore, ownI = bestI, ownJ = bestJ;
        for(int k = 1; k < G; k++) {
            const int source = (g * G + (l + k) % G) & (WAVE - 1);
            const int32_t os = __shfl(ownScore, source, WAVE);
            const int32_t oi = __shfl(ownI, source, WAVE);
            const int32_t oj = __shfl(ownJ, source, WAVE);
            if(os > bestScore || (os == bestScore && DpTie<TIE>::endCellWins(oi, oj, bestI, bestJ))) { bestScore = os; bestI = oi; bestJ = oj; }
        }
    }
    if(hasTask && l == 0) {
        DpEnd e; e.traceOffset = bundleOffsets[bundle]; e.bestI = bestI; e.bestJ = bestJ; e.score = bestScore; e.laneBase = uint32_t(g * G); e.bundleIterations = iters; e.pad = 0;
        ends[t] = e;
    }
}

// Traceback.  ONE LANE per task (64 paths in flight per wavefront) walks from the end cell through the packed trace and
// writes the aligned ordinals.
//
// What the walk costs is latency, not bytes: a launch over sorted tasks lasts as long as its longest path times the
// time of one step.  So:
//  * the trace is consumed in 256-byte chunks (a whole number of iterations of the forward kernel: 8, 4, 2, 1 for C = 2,
//    4, 8, 16 diagonals per lane).  The chunk under the path sits in the lane's private window of LDS, the next one is in
//    flight in registers, its loads issued a whole chunk ahead: a step is a little address arithmetic, ONE LDS read of the
//    two dwords that hold the cell's code, two bit extracts and the move;
//  * the aligned pairs a lane finds in a chunk (at most one per iteration) wait in LDS and are stored together at the
//    start of the next chunk, where the wave waits for the prefetch anyway: a store inside the walk makes the walk wait for
//    the prefetch each time (loads and stores retire through one in-order counter);
//  * ONE kernel for every band class, ONE launch per batch: the launch lasts as long as the longest path of the batch
//    whichever way the classes are split, and three launches (this round's first version: register-resident chunks with
//    the walk unrolled over a chunk's iterations and both cells of each for C = 2 and 4 -- 66 VALU instructions per
//    possible cell, issued whether the lane's path was there or not -- plus this kernel for C = 8, 16) paid for three
//    longest paths: 77 ms per step solo against 52 for this kernel alone, before it was trimmed;
//  * the longest tasks go first (the list is sorted by class, then by ascending length: lane order is reversed), so that
//    the tail of the launch is made of short paths.
//  * (round 3, measured and dropped: ONE window per bundle instead of one per lane -- the lanes of a bundle read the same
//    records, so a wavefront's 16 KB held four chunks of each bundle and the 64 lanes filled it together, a whole epoch of 16
//    iterations ahead; with 8 KB windows four workgroups per CU instead of two; the 64-lane classes in a launch of their own
//    on the side stream; branch-free steps; pairs staged in LDS and stored seven at a time.  Parity-green, 4.9 ms per launch
//    for the narrow classes + 5.4 ms beside it for the wide ones against 4.6 ms for this kernel, 196 against 188 ms per step:
//    neither the memory latency per chunk nor the occupancy is what a launch waits for -- it is its longest paths, at a
//    dependent chain of address arithmetic, one LDS read and the move per step; without any stores the narrow launch still
//    took 3.7 ms.  profiles/r03_traceback_experiments.log.)
// The walk itself only stores the aligned pairs (from the end of the task's ordinal range downwards) and counts
// them; everything that can be computed from the stored pairs afterwards -- AlignmentInfo's metrics, the inner
// acceptance -- is computed by dpMetricsKernel, a wavefront per task, in parallel: each instruction taken out of the
// walk shortens the critical path of the launch (its longest path) by one issue slot per step.
__device__ __forceinline__ void tracebackFinish(uint32_t pos, const PairDesc& pd, const DpEnd& e, uint64_t ordBase, uint32_t t, DpResult* __restrict__ results)
{
    DpResult r;
    r.ordBegin = ordBase + pos;
    r.sumOffset = 0;
    r.markerCount = min(pd.nx, pd.ny) - pos;
    r.first0 = r.first1 = r.last0 = r.last1 = 0; r.minOffset = r.maxOffset = 0; r.maxSkip = r.maxDrift = 0;
    r.passes = 0; r.score = e.score; r.compressedBytes = 0;
    results[t] = r;
}

// A streak of the shasta::compress form (src/compressAlignment.cpp): the skips of its first pair against the pair before and
// its length, in 1, 2, 4, 8 or 16 bytes.
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


// The inner acceptance of a task (src/Align4.cpp:944-981) from its metrics, and its candidate's best component (:132-139): most
// aligned markers; ties resolved towards the component whose first cell in (iY, iX) order comes first, and flagged later.
__device__ __forceinline__ void taskAcceptance(DpResult& r, const PairDesc& pd, const DpTask& task, const DeviceOptions& opt, unsigned long long* __restrict__ pairBest)
{
    const uint32_t count = r.markerCount;
    bool pass = count > 0 && uint64_t(count) >= opt.minAlignedMarkerCount;
    if(pass) {
        const double f0 = double(count) / double(r.last0 + 1 - r.first0);
        const double f1 = double(count) / double(r.last1 + 1 - r.first1);
        if(min(f0, f1) < opt.minAlignedFraction) pass = false;
        if(uint64_t(r.maxSkip) > opt.maxSkip || uint64_t(r.maxDrift) > opt.maxDrift) pass = false;
        const uint32_t leftTrim = min(r.first0, r.first1);
        const uint32_t rightTrim = min(pd.nx - 1 - r.last0, pd.ny - 1 - r.last1);
        if(uint64_t(leftTrim) > opt.maxTrim || uint64_t(rightTrim) > opt.maxTrim) pass = false;
    }
    r.passes = pass ? 1u : 0u;
    if(pass) atomicMax(&pairBest[task.pair], ((unsigned long long)count << 32) | (unsigned long long)(0xffffffffu - task.label));
}

// Tasks [taskBegin, taskEnd) of the sorted list, all of a class with C diagonals per lane (C = 2 or 4).
constexpr int DP_TRACE_CHUNK_QUADS = 16;                    // 16-byte pieces of a 256-byte chunk: {low plane, high plane} of one diagonal of one iteration
// (round 6) From this many iterations on a task's path is walked by a WAVEFRONT of its own (dpTracebackWaveKernel): the lane kernel has one
// chunk of the trace in flight per lane -- one or two iterations of a wide band -- and a launch lasts as long as its longest path at a
// memory latency every chunk: 13 ms for the 30 000-iteration tasks of the ultra-long shape, 172 ms of 700 per step.
constexpr uint32_t DP_TRACEBACK_LONG = 2048;
__global__ void __launch_bounds__(256)
dpTracebackKernel(
    const PairDesc* __restrict__ pairs, const DpTask* __restrict__ tasks, const uint32_t* __restrict__ sortedIds, uint32_t taskBegin, uint32_t taskEnd,
    const DpEnd* __restrict__ ends, const uint64_t* __restrict__ trace,
    const uint64_t* __restrict__ ordOffsets, uint32_t* __restrict__ ordScratch, DpResult* __restrict__ results)
{
    constexpr int QUADS = DP_TRACE_CHUNK_QUADS, MAX_IPC = QUADS / 2;
    __shared__ uint4 window[256 * QUADS];                  // [piece][thread]: conflict-free 16-byte accesses for a wave
    __shared__ uint2 found[256 * MAX_IPC];                 // [k][thread]: the aligned pairs of the current chunk
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= taskEnd - taskBegin) return;
    const uint32_t t = sortedIds[taskEnd - 1 - k];         // longest first
    const DpTask task = tasks[t];
    const PairDesc pd = pairs[task.pair];
    const DpEnd e = ends[t];
    const DpGeometry geo = dpGeometry(task.bandMin, task.bandMax, pd.nx, pd.ny);
    const int cLog2 = dpDiagonalsLog2(geo.cls);
    const uint32_t cMask = (1u << cLog2) - 1u;
    const int ipcLog2 = 4 - cLog2;                         // QUADS / C iterations per chunk
    const uint32_t ipcMask = (1u << ipcLog2) - 1u;
    const uint4* __restrict__ tr = reinterpret_cast<const uint4*>(trace + e.traceOffset);
    const uint32_t* const window32 = reinterpret_cast<const uint32_t*>(window) + 4 * threadIdx.x;
    const uint64_t ordBase = ordOffsets[t];
    uint32_t pos = min(pd.nx, pd.ny);                      // aligned pairs are stored from the end of the task's range downwards
    int32_t i = e.bestI, j = e.bestJ;
    if(geo.iters >= DP_TRACEBACK_LONG) return;             // (a long path: dpTracebackWaveKernel's, result included)
    // Epochs: every lane moves its prefetched chunk into the window and prefetches the next one at
    // the same point of the program, then walks until its path leaves the chunk.  The wave waits
    // for memory once per epoch, for loads issued a whole epoch earlier.
    bool active = e.score > NEG_SCORE && i > 0 && j > 0;
    // The tasks of a bundle share their trace records, and they sit in adjacent lanes here: every one of them counts its
    // chunks down from the BUNDLE's last chunk, so that the lanes of a bundle ask for the same 256 bytes in the same
    // instruction (one request) -- each starting at its own last chunk, an epoch or two apart, made the 4 tasks of a bundle
    // fetch the trace 2.3 times over (PMC: 19.6 GB per launch for 8.4 GB of trace).  A lane whose path starts lower waits.
    const int32_t ownChunk = active ? int32_t((uint32_t(i + j - geo.s0) >> 1) >> ipcLog2) : -1;
    int32_t chunk = active ? int32_t((e.bundleIterations - 1u) >> ipcLog2) : -1;
    uint4 next[QUADS];
#pragma unroll
    for(int q = 0; q < QUADS; q++) next[q] = (active && chunk == ownChunk) ? tr[int64_t(chunk) * QUADS + q] : make_uint4(0, 0, 0, 0);
    uint32_t foundCount = 0;
    auto storeFound = [&]() {
        for(uint32_t q = 0; q < foundCount; q++) { --pos; *reinterpret_cast<uint2*>(ordScratch + 2 * (ordBase + pos)) = found[q * 256 + threadIdx.x]; }
        foundCount = 0;
    };
    while(__any(active)) {
        // The pairs of the previous chunk leave here, where the wave waits for the prefetched chunk anyway (stores and
        // loads retire through one in-order counter: stores issued after the prefetch would wait for it).
        storeFound();
        if(active && chunk <= ownChunk) {
#pragma unroll
            for(int q = 0; q < QUADS; q++) window[q * 256 + threadIdx.x] = next[q];
        }
        if(active && chunk - 1 <= ownChunk && chunk > 0) {
#pragma unroll
            for(int q = 0; q < QUADS; q++) next[q] = tr[int64_t(chunk - 1) * QUADS + q];
        }
        while(active && chunk <= ownChunk) {
            const uint32_t b = uint32_t(i - j - task.bandMin);
            const uint32_t it = uint32_t(i + j - geo.s0) >> 1;
            if(int32_t(it >> ipcLog2) != chunk) break;
            const uint32_t bit = e.laneBase + (b >> cLog2);
            const uint32_t piece = ((it & ipcMask) << cLog2) + (b & cMask);
            // The piece is {low plane: bits 0-31, 32-63; high plane: bits 0-31, 32-63}: the two dwords that hold bit `bit`.
            const uint32_t* const word = window32 + piece * 1024u + ((bit >> 5) & 1u);
            const uint32_t lo = word[0], hi = word[2];
            const uint32_t dir = ((lo >> (bit & 31u)) & 1u) | (((hi >> (bit & 31u)) & 1u) << 1);
            // A diagonal step over equal kmers: an aligned marker pair (src/Align4.cpp:1057-1061).
            if(dir == 0u) { found[foundCount * 256 + threadIdx.x] = make_uint2(uint32_t(i - 1), uint32_t(j - 1)); ++foundCount; }
            i -= (dir != 2u) ? 1 : 0;
            j -= (dir != 3u) ? 1 : 0;
            active = i > 0 && j > 0;
        }
        --chunk;
    }
    storeFound();
    tracebackFinish(pos, pd, e, ordBase, t, results);
}

// The walk of a LONG path (DP_TRACEBACK_LONG iterations and more): one wavefront per task.  The trace goes through LDS in blocks of 32
// chunks (8 KB), the block under the path in one half of a ring, the block below it asked for -- 64 lanes x 8 loads, all in flight at once --
// when the walk enters the one above: by the time the walk gets there the data has been in registers for a whole block.  Every lane walks (the
// same LDS words for all: broadcast reads), lane 0 notes the aligned pairs in LDS, the wavefront stores them together.  Same decisions, same
// output as dpTracebackKernel, which leaves these tasks alone.
constexpr int DP_WAVE_BLOCK_CHUNKS = 32, DP_WAVE_FOUND = 512;
__global__ void __launch_bounds__(64)
dpTracebackWaveKernel(
    const PairDesc* __restrict__ pairs, const DpTask* __restrict__ tasks, const uint32_t* __restrict__ sortedIds, uint32_t taskCount,
    const DpEnd* __restrict__ ends, const uint64_t* __restrict__ trace,
    const uint64_t* __restrict__ ordOffsets, uint32_t* __restrict__ ordScratch, DpResult* __restrict__ results)
{
    constexpr int QUADS = DP_TRACE_CHUNK_QUADS, BLOCK_QUADS = DP_WAVE_BLOCK_CHUNKS * QUADS, PER_LANE = BLOCK_QUADS / WAVE;
    __shared__ uint4 ring[2][BLOCK_QUADS];
    __shared__ uint2 found[DP_WAVE_FOUND];
    const uint32_t k = blockIdx.x;
    if(k >= taskCount) return;
    const int lane = laneId();
    const uint32_t t = sortedIds[taskCount - 1 - k];
    const DpTask task = tasks[t];
    const PairDesc pd = pairs[task.pair];
    const DpGeometry geo = dpGeometry(task.bandMin, task.bandMax, pd.nx, pd.ny);
    if(geo.iters < DP_TRACEBACK_LONG) return;              // (dpTracebackKernel's)
    const DpEnd e = ends[t];
    const int cLog2 = dpDiagonalsLog2(geo.cls);
    const uint32_t cMask = (1u << cLog2) - 1u;
    const int ipcLog2 = 4 - cLog2;                         // QUADS / C iterations per chunk
    const uint32_t ipcMask = (1u << ipcLog2) - 1u;
    const uint4* __restrict__ tr = reinterpret_cast<const uint4*>(trace + e.traceOffset);
    const int64_t lastQuad = (int64_t((e.bundleIterations - 1u) >> ipcLog2) + 1) * QUADS - 1;       // the bundle's trace ends here
    const uint64_t ordBase = ordOffsets[t];
    uint32_t pos = min(pd.nx, pd.ny);                      // aligned pairs are stored from the end of the task's range downwards
    int32_t i = e.bestI, j = e.bestJ;
    uint32_t foundCount = 0;
    auto storeFound = [&]() {
        waveLdsSync();
        for(uint32_t q = uint32_t(lane); q < foundCount; q += WAVE) *reinterpret_cast<uint2*>(ordScratch + 2 * (ordBase + pos - 1u - q)) = found[q];
        pos -= foundCount; foundCount = 0;
        waveLdsSync();
    };
    if(e.score > NEG_SCORE && i > 0 && j > 0) {
        auto blockOf = [&](int32_t ii, int32_t jj) { return int32_t(((uint32_t(ii + jj - geo.s0) >> 1) >> ipcLog2) / uint32_t(DP_WAVE_BLOCK_CHUNKS)); };
        uint4 ahead[PER_LANE];
        auto loadBlock = [&](int32_t block) {
#pragma unroll
            for(int a = 0; a < PER_LANE; a++) ahead[a] = tr[min(int64_t(block) * BLOCK_QUADS + a * WAVE + lane, lastQuad)];
        };
        auto keepBlock = [&](int32_t block) {
#pragma unroll
            for(int a = 0; a < PER_LANE; a++) ring[block & 1][a * WAVE + lane] = ahead[a];
        };
        int32_t block = blockOf(i, j);
        loadBlock(block); keepBlock(block);
        if(block > 0) loadBlock(block - 1);
        waveLdsSync();
        for(;;) {
            // The walk inside the block.
            const uint32_t* const words = reinterpret_cast<const uint32_t*>(ring[block & 1]);
            bool done = false;
            for(;;) {
                const uint32_t b = uint32_t(i - j - task.bandMin);
                const uint32_t it = uint32_t(i + j - geo.s0) >> 1;
                const uint32_t chunk = it >> ipcLog2;
                if(int32_t(chunk / uint32_t(DP_WAVE_BLOCK_CHUNKS)) != block) break;
                const uint32_t bit = e.laneBase + (b >> cLog2);
                const uint32_t piece = (chunk % uint32_t(DP_WAVE_BLOCK_CHUNKS)) * QUADS + ((it & ipcMask) << cLog2) + (b & cMask);
                // The piece is {low plane: bits 0-31, 32-63; high plane: bits 0-31, 32-63}: the two dwords that hold bit `bit`.
                const uint32_t lo = words[4u * piece + ((bit >> 5) & 1u)], hi = words[4u * piece + 2u + ((bit >> 5) & 1u)];
                const uint32_t dir = ((lo >> (bit & 31u)) & 1u) | (((hi >> (bit & 31u)) & 1u) << 1);
                // A diagonal step over equal kmers: an aligned marker pair (src/Align4.cpp:1057-1061).
                if(dir == 0u) {
                    if(lane == 0) found[foundCount] = make_uint2(uint32_t(i - 1), uint32_t(j - 1));
                    if(++foundCount == uint32_t(DP_WAVE_FOUND)) storeFound();
                }
                i -= (dir != 2u) ? 1 : 0;
                j -= (dir != 3u) ? 1 : 0;
                if(!(i > 0 && j > 0)) { done = true; break; }
            }
            if(done || block == 0) break;
            // Into the block below: what was asked for a block ago goes into the other half of the ring, the block below that is asked for.
            --block;
            keepBlock(block);
            if(block > 0) loadBlock(block - 1);
            waveLdsSync();
        }
    }
    storeFound();
    if(lane == 0) tracebackFinish(pos, pd, e, ordBase, t, results);
}
