// Align4 on MI355X, K10a: the banded overlap alignment of a task that has SEVERAL optimal chains of matches (sparseChainKernel,
// align4_sparse.hpp, leaves such a task SPARSE_AMBIGUOUS), with the dense DP confined to where the chains differ.  Included by
// align4.hip inside its anonymous namespace, after align4_sparse.hpp.
//
// What it computes (oracle/anchored_chain.hpp states it on the CPU in the same steps, and tests it against the dense DP under all
// twelve tie policies): a hit that lies on EVERY optimal chain -- an anchor -- is in the dense traceback's output whatever the tie
// policy; between two consecutive anchors either no other hit lies on an optimal chain (nothing to add there), or several
// sub-chains tie, and then the policy decides: the dense DP runs on the rectangle between the two anchors alone, its corners
// fixed (paths start in the cell behind the first anchor, the traceback starts in the cell in front of the second; before the
// first and after the last anchor the free border stays what it is).  Every cell of the path the whole-matrix traceback walks
// lies on an optimal path, all of which pass both anchors, so the cell's tying predecessors are inside the rectangle with the
// same values relative to the first anchor, and the others can only lose more: the same decisions, cell by cell.
// At 100 k reads the 13 % of the tasks that reach this kernel are left with 0.15 % of the dense path's cells
// (profiles/r04_sparse_census.txt), in rectangles of a few hundred cells.
//
// A WAVEFRONT per task, tasks taken from the list sparseChainKernel appended them to (longest first, about):
//   sweep      from the last hit back, 64 link words at a time (one coalesced read), the 64 steps on the scalar unit: a hit is live
//              if it ends an optimal chain or a live later hit links to it; it is an anchor if, when the sweep reaches it, no link of
//              a live later hit passes over it (to an earlier hit, or to the border) and no optimal chain ends before it
//   windows    maximal runs of live hits that are not anchors, each between two anchors (or an anchor and the border): up to
//              ANCHOR_MAX_WINDOWS of them, found on the scalar unit from the bit words of the sweep
//   rectangle  row by row, the lanes across the row: H(i, j) = max(c(j), H(i, j - 1) - 1) is a prefix maximum of c(j) + j; the moves
//              (and whether the markers are equal) in a byte per cell in LDS, the traceback read from there
//   emit       anchors and the rectangles' pairs from the end of the task's range of the ordinal scratch downwards, 64 hits per step
// A task with no anchor, a live hit whose optimal links reach further back than its link word names (29 hits), a rectangle of more than ANCHOR_MAX_CELLS cells or ANCHOR_MAX_SIDE markers on a side, too many windows or
// too many pairs inside them stays for the dense kernels (state SPARSE_DENSE), as before.
#pragma once
// (An emulated build compiled with -D'ANCHOR_REASON(why, n)=...' reports why the anchor kernel left a task to the dense kernels:
// 1 no anchor, 2 too many windows, 3 a rectangle or its pairs beyond the limits, 4 an optimal link of a live hit beyond its link word.)
#ifndef ANCHOR_REASON
#define ANCHOR_REASON(why, n)
#endif

constexpr int ANCHOR_MAX_BITWORDS = (2 * 8192 + 64 + 63) / 64;     // of a task's hits: 16 448 of them (the wave kernel's largest class holds 15 360; the lane kernel lists up to sparseListCapacity: a task with more stays for the dense kernels)
constexpr int ANCHOR_MAX_WINDOWS = 128;
constexpr int ANCHOR_MAX_CELLS = 4096;           // (wx + 1)(wy + 1) of a rectangle: a byte each (census, profiles/r04_sparse_census.txt: 4 of 2 589 tasks have a larger one; 22 KB of LDS per wavefront instead of 37: seven workgroups per CU instead of four)
constexpr int ANCHOR_MAX_SIDE = 511;             // markers on a side of a rectangle
constexpr int ANCHOR_MAX_PAIRS = 512;            // aligned pairs inside the windows of one task
constexpr int32_t ANCHOR_NEG = -(1 << 28);
static_assert(GAP_SCORE == -1, "the row's prefix maximum of c(j) + j is the recurrence for a gap of -1");
constexpr uint32_t ANCHOR_GRID = 4096;           // workgroups of one wavefront; each takes tasks blockIdx.x, + gridDim.x, ...

// What a wavefront holds of its task.  Two sizes: the first launch's (22 KB: seven workgroups per CU) and, for the few tasks with a
// rectangle beyond it (0.3 % of the tasks at 100 k reads: they went to the dense kernels, a handful of wavefronts per launch, 60 ms of
// launches per step for 0.1 % of the DP cells), a second launch's (118 KB, dynamic LDS, one workgroup per CU).
template<int MAX_CELLS, int MAX_SIDE, int MAX_PAIRS, bool BAND_TRACE = false>
struct AnchorSharedT {
    static constexpr int maxCells = MAX_CELLS, maxSide = MAX_SIDE, maxPairs = MAX_PAIRS;
    static constexpr bool bandTrace = BAND_TRACE;        // the trace holds the band's cells only, two bits each (anchorRectangleBand)
    uint64_t live[ANCHOR_MAX_BITWORDS], anchor[ANCHOR_MAX_BITWORDS];
    uint8_t trace[MAX_CELLS];                    // [i * (wy + 1) + j]: move (DpTie::Move) | markers equal << 2
    int32_t row[2][MAX_SIDE + 1];                // H of the previous and of the current row
    int32_t lastColumn[MAX_SIDE + 1];            // H(i, wy)
    uint32_t kmers0[MAX_SIDE + 1], kmers1[MAX_SIDE + 1];
    uint32_t pairs[MAX_PAIRS];                   // x << 16 | y, every window's from its last pair to its first
    int32_t windowFrom[ANCHOR_MAX_WINDOWS], windowTo[ANCHOR_MAX_WINDOWS];       // anchors (hit indices) before and behind; -1 / n: the border
    uint32_t windowPairs[ANCHOR_MAX_WINDOWS], windowBegin[ANCHOR_MAX_WINDOWS];
};
using AnchorShared = AnchorSharedT<ANCHOR_MAX_CELLS, ANCHOR_MAX_SIDE, ANCHOR_MAX_PAIRS>;
// The second launch: 96 KB of trace = 393 216 cells of the BAND at two bits (a rectangle that is most of a task -- a task with hardly any
// anchor -- has a million cells, of which the band holds a tenth), 8 192 pairs, and (round 6) sides of ANY length: the rows of H hold the
// band's cells only (position = column - the band's lowest column in that row: 1 024 at most), the markers of read 1 under the band
// slide through a ring, those of read 0 come a block of 64 rows at a time in a register, the last column's values are kept for the rows
// whose band reaches it.  (Until then the rows were as long as the rectangle's side, 3 071 markers at most by the LDS: the one reason
// left for a task of the 100 k-read workload to end in the dense kernels -- 26 a step, 16 ms of launches.)
constexpr int ANCHOR_BIG_TRACE_BYTES = 96 * 1024, ANCHOR_BIG_BAND = 1024, ANCHOR_BIG_PAIRS = 8192, ANCHOR_BIG_RING = 2048;
struct AnchorSharedBig {
    static constexpr int maxCells = 4 * ANCHOR_BIG_TRACE_BYTES, maxSide = 0x7ffffff0, maxPairs = ANCHOR_BIG_PAIRS;
    static constexpr bool bandTrace = true;
    uint64_t live[ANCHOR_MAX_BITWORDS], anchor[ANCHOR_MAX_BITWORDS];
    uint8_t trace[ANCHOR_BIG_TRACE_BYTES];       // two bits per cell of the band: [row * rowWords + position / 16]; 0 diagonal over different markers, 1 vertical, 2 horizontal, 3 diagonal over equal markers
    int32_t row[2][ANCHOR_BIG_BAND + 2];         // H of the previous and of the current row, by position in the band (one more: "outside")
    int32_t lastColumn[ANCHOR_BIG_BAND + 2];     // H(i, wy) of the rows whose band holds the last column, from the first of them on
    uint32_t ring1[ANCHOR_BIG_RING];             // markers of read 1, [index & (ANCHOR_BIG_RING - 1)]: those under the band of the rows being computed
    uint32_t pairs[ANCHOR_BIG_PAIRS];
    int32_t windowFrom[ANCHOR_MAX_WINDOWS], windowTo[ANCHOR_MAX_WINDOWS];
    uint32_t windowPairs[ANCHOR_MAX_WINDOWS], windowBegin[ANCHOR_MAX_WINDOWS];
};
static_assert(sizeof(AnchorSharedBig) <= 160 * 1024, "the second launch's LDS");

__device__ __forceinline__ uint64_t bitsFrom(int b) { return b >= 64 ? 0ULL : ~0ULL << b; }       // bits b .. 63
__device__ __forceinline__ int32_t waveMaxScan(int32_t v, int lane)                               // inclusive prefix maximum over the lanes
{
#pragma unroll
    for(int d = 1; d < WAVE; d <<= 1) { const int32_t o = __shfl_up(v, d, WAVE); if(lane >= d) v = max(v, o); }
    return v;
}

// The rectangle of markers [x0, x1] x [y0, y1]; the aligned pairs of its traced path go to sh.pairs[begin ...] from the last to the
// first.  Returns their number, or -1 if they do not fit.
template<int TIE, class Shared>
__device__ int32_t anchorRectangle(Shared& sh, const uint32_t* __restrict__ p0, const uint32_t* __restrict__ p1,
    int32_t x0, int32_t x1, int32_t y0, int32_t y1, bool beginFixed, bool endFixed, int32_t bandMin, int32_t bandMax, uint32_t begin, int lane)
{
    using Tie = DpTie<TIE>;
    const int32_t wx = x1 - x0 + 1, wy = y1 - y0 + 1, stride = wy + 1;
    for(int32_t a = lane; a < wx; a += WAVE) sh.kmers0[a] = p0[x0 + a];
    for(int32_t a = lane; a < wy; a += WAVE) sh.kmers1[a] = p1[y0 + a];
    // Row i holds columns [jLo(i), jHi(i)] of the band.
    const int32_t shift = x0 - y0;
    auto jLow = [&](int32_t i) { return max(0, i + shift - bandMax); };
    auto jHigh = [&](int32_t i) { return min(wy, i + shift - bandMin); };
    {
        const int32_t lo = jLow(0), hi = jHigh(0);
        for(int32_t j = lane; j <= wy; j += WAVE) {
            // (fixed corner: the border is reached from the corner by gaps, as long as the band holds it; the corner's own
            // diagonal is the first anchor's, inside the band)
            const bool in = j >= lo && j <= hi;
            sh.row[0][j] = !in ? ANCHOR_NEG : (beginFixed ? -j : 0);
            sh.trace[j] = uint8_t(Tie::VERTICAL);
        }
        if(lane == 0) sh.lastColumn[0] = (wy >= lo && wy <= hi) ? (beginFixed ? -wy : 0) : ANCHOR_NEG;
    }
    waveLdsSync();
    for(int32_t i = 1; i <= wx; i++) {
        const int32_t* const previous = sh.row[(i - 1) & 1];
        int32_t* const current = sh.row[i & 1];
        const int32_t lo = jLow(i), hi = jHigh(i);
        const uint32_t k0 = sh.kmers0[i - 1];
        int32_t carry = 2 * ANCHOR_NEG;                     // the largest c(j') + j' of the chunks before
        for(int32_t jBase = 0; jBase <= wy; jBase += WAVE) {
            const int32_t j = jBase + lane;
            const bool in = j >= lo && j <= hi;
            int32_t diagonal = ANCHOR_NEG, horizontal = ANCHOR_NEG;
            bool equal = false;
            if(in && j <= wy) {
                const int32_t h0 = previous[j];
                horizontal = h0 <= ANCHOR_NEG ? ANCHOR_NEG : h0 - 1;
                if(j > 0) {
                    const int32_t d0 = previous[j - 1];
                    equal = k0 == sh.kmers1[j - 1];
                    diagonal = d0 <= ANCHOR_NEG ? ANCHOR_NEG : d0 + (equal ? MATCH_SCORE : MISMATCH_SCORE);
                }
                else if(!beginFixed) horizontal = 0;        // the free border: H(i, 0) = 0
            }
            const int32_t c = max(diagonal, horizontal);
            const int32_t scanned = max(waveMaxScan(c + j, lane), carry);
            carry = __shfl(scanned, WAVE - 1, WAVE);
            int32_t h = scanned - j;
            if(!in || j > wy || h <= ANCHOR_NEG / 2) h = ANCHOR_NEG;
            // The move, by the policy's order among the candidates that attain h.
            int32_t left = __shfl_up(h, 1, WAVE);            // H(i, j - 1)
            if(lane == 0) left = jBase ? current[jBase - 1] : ANCHOR_NEG;
            const int32_t vertical = (j == 0 || left <= ANCHOR_NEG) ? ANCHOR_NEG : left - 1;
            int move;
            if(j == 0) move = Tie::HORIZONTAL;
            else {
                bool attains[3];
                attains[Tie::DIAGONAL] = diagonal == h; attains[Tie::VERTICAL] = vertical == h; attains[Tie::HORIZONTAL] = horizontal == h;
                move = attains[Tie::first] ? Tie::first : (attains[Tie::second] ? Tie::second : Tie::third);
            }
            if(j <= wy) {
                current[j] = h;
                sh.trace[i * stride + j] = uint8_t(move | (equal ? 4 : 0));
                if(j == wy) sh.lastColumn[i] = h;
            }
            waveLdsSync();                                   // (lane 0 of the next chunk reads current[jBase - 1])
        }
    }
    // Where the traceback starts.  (A fixed end that no path from the fixed begin reaches, or a walk that does not end in the fixed
    // begin, would contradict what the anchors are: the task goes to the dense kernels rather than on.)
    int32_t i = wx, j = wy;
    if(endFixed && sh.row[wx & 1][wy] <= ANCHOR_NEG) return -2;                  // (-2: the walk contradicts the anchors)
    if(!endFixed) {
        // The border cells in the order the dense DP scans them -- (0, wy) ... (wx - 1, wy), then (wx, 0) ... (wx, wy) -- and the
        // first or the last of those with the largest score.
        const int32_t* const last = sh.row[wx & 1];
        const int32_t total = wx + wy + 1;
        int32_t bestScore = ANCHOR_NEG, bestAt = Tie::lastMaximum ? -1 : 0x7fffffff;
        for(int32_t a = lane; a < total; a += WAVE) {
            const int32_t score = a < wx ? sh.lastColumn[a] : last[a - wx];
            if(score <= ANCHOR_NEG) continue;
            if(score > bestScore || (score == bestScore && (Tie::lastMaximum ? a > bestAt : a < bestAt))) { bestScore = score; bestAt = a; }
        }
#pragma unroll
        for(int d = 32; d >= 1; d >>= 1) {
            const int32_t otherScore = __shfl_xor(bestScore, d, WAVE), otherAt = __shfl_xor(bestAt, d, WAVE);
            if(otherScore > bestScore || (otherScore == bestScore && otherScore > ANCHOR_NEG && (Tie::lastMaximum ? otherAt > bestAt : otherAt < bestAt))) { bestScore = otherScore; bestAt = otherAt; }
        }
        if(bestScore <= ANCHOR_NEG) return 0;
        if(bestAt < wx) { i = bestAt; j = wy; } else { i = wx; j = bestAt - wx; }
    }
    // The traceback: the same walk in every lane (LDS reads of one address), lane 0 writes.
    int32_t found = 0;
    for(int32_t steps = 0; steps <= wx + wy + 1; steps++) {
        if(beginFixed ? (i == 0 && j == 0) : (i == 0 || j == 0)) break;
        const uint32_t t = sh.trace[i * stride + j];
        const int move = int(t & 3u);
        if(move == Tie::DIAGONAL) {
            --i; --j;
            if(t & 4u) {
                if(begin + uint32_t(found) >= uint32_t(Shared::maxPairs)) return -1;        // (-1: the pairs do not fit)
                if(lane == 0) sh.pairs[begin + uint32_t(found)] = (uint32_t(x0 + i) << 16) | uint32_t(y0 + j);
                ++found;
            }
        }
        else if(move == Tie::VERTICAL) --j;
        else --i;
    }
    waveLdsSync();
    if(beginFixed && (i != 0 || j != 0)) return -2;
    return found;
}

// The same rectangle for the second launch: only the cells inside the band are computed, a row's cells at their POSITION in the band
// (position b of row i = column b + i + shift - bandMax: the cell above it has position b + 1 in its row, the one above and to the left
// position b), and only they are traced, two bits each; a diagonal step over equal markers has a code of its own, so the walk reads
// nothing but the trace.  Returns the number of aligned pairs, -1 (the pairs do not fit), -2 (the walk contradicts the anchors) or -3
// (the band's cells do not fit the trace, or the band is wider than the rows).
template<int TIE, class Shared>
__device__ int32_t anchorRectangleBand(Shared& sh, const uint32_t* __restrict__ p0, const uint32_t* __restrict__ p1,
    int32_t x0, int32_t x1, int32_t y0, int32_t y1, bool beginFixed, bool endFixed, int32_t bandMin, int32_t bandMax, uint32_t begin, int lane)
{
    using Tie = DpTie<TIE>;
    constexpr int EQUAL_DIAGONAL = 3;
    static_assert(Tie::DIAGONAL == 0 && Tie::VERTICAL == 1 && Tie::HORIZONTAL == 2, "the fourth code is free");
    const int32_t wx = x1 - x0 + 1, wy = y1 - y0 + 1;
    const int32_t bandWidth = bandMax - bandMin + 1, rowWords = (bandWidth + 15) / 16;
    uint32_t* const traceWords = reinterpret_cast<uint32_t*>(sh.trace);
    if(bandWidth > ANCHOR_BIG_BAND || int64_t(wx + 1) * rowWords > int64_t(Shared::maxCells / 16)) return -3;
    const int32_t shift = x0 - y0;
    auto bandBase = [&](int32_t i) { return i + shift - bandMax; };                  // the column of the band's lowest diagonal in row i (may be negative)
    auto jLow = [&](int32_t i) { return max(0, bandBase(i)); };
    auto jHigh = [&](int32_t i) { return min(wy, i + shift - bandMin); };
    // The codes of the 64 positions a chunk's lanes hold, as the four words of the row's trace they make: two ballots, the planes' bits
    // interleaved by the first four lanes, four plain stores -- every word of a row is written exactly once, so the trace needs no clearing.
    // (Until round 6 every lane OR-ed its two bits in with an LDS atomic: sixteen lanes on each of four words, sixteen serialised atomics per
    // word and chunk.)
    auto spread16 = [](uint32_t x) { x = (x | (x << 8)) & 0x00ff00ffu; x = (x | (x << 4)) & 0x0f0f0f0fu; x = (x | (x << 2)) & 0x33333333u; return (x | (x << 1)) & 0x55555555u; };
    auto putCodes = [&](int32_t i, int32_t chunk, int code) {
        const uint64_t low = __ballot((code & 1) != 0), high = __ballot((code & 2) != 0);
        const int32_t word = chunk * 4 + lane;
        if(lane < 4 && word < rowWords)
            traceWords[i * rowWords + word] = spread16(uint32_t(low >> (16 * lane)) & 0xffffu) | (spread16(uint32_t(high >> (16 * lane)) & 0xffffu) << 1);
    };
    // The rows whose band holds the last column: iFirst .. iFirst + bandWidth - 1.
    const int32_t iFirst = max(0, wy - shift + bandMin);
    for(int32_t a = lane; a < ANCHOR_BIG_BAND + 2; a += WAVE) sh.lastColumn[a] = ANCHOR_NEG;
    // Markers of read 1 (column j compares kmers1[j - 1]) into the ring: what the first 128 rows can touch now, 64 more at the head of
    // every block of 64 rows, loaded a block ahead.
    auto ringStore = [&](int32_t index, uint32_t value) { sh.ring1[uint32_t(index) & uint32_t(ANCHOR_BIG_RING - 1)] = value; };
    int32_t ringLoaded = max(0, jLow(1) - 1);                                    // markers [.., ringLoaded) are in the ring (or below every row's band)
    {
        const int32_t until = min(wy, jHigh(min(wx, 128)));
        for(int32_t a = ringLoaded + lane; a < until; a += WAVE) ringStore(a, p1[y0 + a]);
        ringLoaded = max(ringLoaded, until);
    }
    int32_t pendingIndex = -1;
    uint32_t pendingValue = 0;
    // Markers of read 0: the block of 64 rows in hand in a register, the next one on its way.
    uint32_t block0 = p0[x0 + min(lane, wx - 1)], block0Next = p0[x0 + min(WAVE + lane, wx - 1)];
    waveLdsSync();
    const int32_t chunks = (bandWidth + WAVE) / WAVE;                             // (position bandWidth included: it is written "outside" for the row below)
    for(int32_t chunk = 0; chunk < chunks; chunk++) {
        const int32_t b = chunk * WAVE + lane, j = b + bandBase(0);
        const bool in = b < bandWidth && j >= 0 && j <= wy;
        if(b <= bandWidth) sh.row[0][b] = !in ? ANCHOR_NEG : (beginFixed ? -j : 0);
        putCodes(0, chunk, in ? int(Tie::VERTICAL) : 0);
        if(in && j == wy && iFirst == 0) sh.lastColumn[0] = beginFixed ? -wy : 0;
    }
    waveLdsSync();
    for(int32_t i = 1; i <= wx; i++) {
        if(((i - 1) & (WAVE - 1)) == 0 && i > 1) {
            // A new block of 64 rows: its markers of read 0 become the block in hand, the ring takes what was asked for a block ago, and
            // the next 64 markers of either read are asked for.
            block0 = block0Next;
            block0Next = p0[x0 + min(i - 1 + WAVE + lane, wx - 1)];
            if(pendingIndex >= 0) ringStore(pendingIndex, pendingValue);
            pendingIndex = (ringLoaded + lane < wy) ? ringLoaded + lane : -1;
            pendingValue = p1[y0 + min(ringLoaded + lane, wy - 1)];
            ringLoaded = min(wy, ringLoaded + WAVE);
            waveLdsSync();
        }
        const int32_t* const previous = sh.row[(i - 1) & 1];
        int32_t* const current = sh.row[i & 1];
        const int32_t lo = jLow(i), hi = jHigh(i), base = bandBase(i);
        const uint32_t k0 = __builtin_amdgcn_readlane(block0, (i - 1) & (WAVE - 1));
        int32_t carry = 2 * ANCHOR_NEG;                     // the largest c(j') + j' of the chunks before
        for(int32_t chunk = 0; chunk < chunks; chunk++) {
            const int32_t b = chunk * WAVE + lane, j = b + base;
            const bool in = b < bandWidth && j >= lo && j <= hi;
            int32_t diagonal = ANCHOR_NEG, horizontal = ANCHOR_NEG;
            bool equal = false;
            if(in) {
                const int32_t h0 = previous[b + 1];                                 // H(i - 1, j)
                horizontal = h0 <= ANCHOR_NEG ? ANCHOR_NEG : h0 - 1;
                if(j > 0) {
                    const int32_t d0 = previous[b];                                 // H(i - 1, j - 1)
                    equal = k0 == sh.ring1[uint32_t(j - 1) & uint32_t(ANCHOR_BIG_RING - 1)];
                    diagonal = d0 <= ANCHOR_NEG ? ANCHOR_NEG : d0 + (equal ? MATCH_SCORE : MISMATCH_SCORE);
                }
                else if(!beginFixed) horizontal = 0;        // the free border: H(i, 0) = 0
            }
            const int32_t c = max(diagonal, horizontal);
            const int32_t scanned = max(waveMaxScan(c + j, lane), carry);
            carry = __shfl(scanned, WAVE - 1, WAVE);
            int32_t h = scanned - j;
            if(!in || h <= ANCHOR_NEG / 2) h = ANCHOR_NEG;
            int32_t left = __shfl_up(h, 1, WAVE);            // H(i, j - 1)
            if(lane == 0) left = chunk > 0 ? current[b - 1] : ANCHOR_NEG;          // (before the band's first position the row is outside)
            const int32_t vertical = (j == 0 || left <= ANCHOR_NEG) ? ANCHOR_NEG : left - 1;
            int move;
            if(j == 0) move = Tie::HORIZONTAL;
            else {
                bool attains[3];
                attains[Tie::DIAGONAL] = diagonal == h; attains[Tie::VERTICAL] = vertical == h; attains[Tie::HORIZONTAL] = horizontal == h;
                move = attains[Tie::first] ? Tie::first : (attains[Tie::second] ? Tie::second : Tie::third);
            }
            if(b <= bandWidth) {
                current[b] = h;
                if(in && j == wy && i >= iFirst && i - iFirst <= ANCHOR_BIG_BAND + 1) sh.lastColumn[i - iFirst] = h;
            }
            putCodes(i, chunk, !in ? 0 : ((move == Tie::DIAGONAL && equal) ? EQUAL_DIAGONAL : move));
            waveLdsSync();                                   // (lane 0 of the next chunk reads current[b - 1])
        }
    }
    waveLdsSync();                                           // (the end cells are read by all lanes)
    int32_t i = wx, j = wy;
    const bool cornerInBand = wy >= jLow(wx) && wy <= jHigh(wx);
    const int32_t* const last = sh.row[wx & 1];
    if(endFixed && (!cornerInBand || last[wy - bandBase(wx)] <= ANCHOR_NEG)) {
#ifdef ANCHOR_DEBUG
        if(lane == 0) std::fprintf(stderr, "band rectangle: end corner unreachable: wx %d wy %d shift %d band [%d, %d] cornerInBand %d beginFixed %d\n", wx, wy, shift, bandMin, bandMax, int(cornerInBand), int(beginFixed));
#endif
        return -2;
    }
    if(!endFixed) {
        // The border cells in the order the dense DP scans them -- (0, wy) ... (wx - 1, wy), then (wx, 0) ... (wx, wy) -- and the first or
        // the last of those with the largest score: the last column's cells the band holds, then the last row's.
        const int32_t lastLo = jLow(wx), lastHi = jHigh(wx);
        int32_t bestScore = ANCHOR_NEG, bestAt = Tie::lastMaximum ? -1 : 0x7fffffff;
        auto take = [&](int32_t score, int32_t a) {
            if(score <= ANCHOR_NEG) return;
            if(score > bestScore || (score == bestScore && (Tie::lastMaximum ? a > bestAt : a < bestAt))) { bestScore = score; bestAt = a; }
        };
        for(int32_t a = iFirst + lane; a < wx && a - iFirst <= ANCHOR_BIG_BAND + 1; a += WAVE) take(sh.lastColumn[a - iFirst], a);
        for(int32_t column = lastLo + lane; column <= lastHi; column += WAVE) take(last[column - bandBase(wx)], wx + column);
#pragma unroll
        for(int d = 32; d >= 1; d >>= 1) {
            const int32_t otherScore = __shfl_xor(bestScore, d, WAVE), otherAt = __shfl_xor(bestAt, d, WAVE);
            if(otherScore > bestScore || (otherScore == bestScore && otherScore > ANCHOR_NEG && (Tie::lastMaximum ? otherAt > bestAt : otherAt < bestAt))) { bestScore = otherScore; bestAt = otherAt; }
        }
        if(bestScore <= ANCHOR_NEG) return 0;
        if(bestAt < wx) { i = bestAt; j = wy; } else { i = wx; j = bestAt - wx; }
    }
    // The traceback: the same walk in every lane (LDS reads of one address), lane 0 writes.
    int32_t found = 0;
    for(int32_t steps = 0; steps <= wx + wy + 1; steps++) {
        if(beginFixed ? (i == 0 && j == 0) : (i == 0 || j == 0)) break;
        const int32_t b = j - bandBase(i);
        if(b < 0 || b >= bandWidth) {
#ifdef ANCHOR_DEBUG
            if(lane == 0) std::fprintf(stderr, "band rectangle: walk left the band at (%d, %d) b %d after %d steps: wx %d wy %d shift %d band [%d, %d] beginFixed %d endFixed %d x0 %d y0 %d\n", i, j, b, steps, wx, wy, shift, bandMin, bandMax, int(beginFixed), int(endFixed), x0, y0);
#endif
            return -2;               // (the walk left the band: it cannot)
        }
        const int move = int((traceWords[i * rowWords + (b >> 4)] >> (2 * (b & 15))) & 3u);
        if(move == Tie::DIAGONAL || move == EQUAL_DIAGONAL) {
            --i; --j;
            if(move == EQUAL_DIAGONAL) {
                if(begin + uint32_t(found) >= uint32_t(Shared::maxPairs)) return -1;
                if(lane == 0) sh.pairs[begin + uint32_t(found)] = (uint32_t(x0 + i) << 16) | uint32_t(y0 + j);
                ++found;
            }
        }
        else if(move == Tie::VERTICAL) --j;
        else --i;
    }
    waveLdsSync();
    if(beginFixed && (i != 0 || j != 0)) {
#ifdef ANCHOR_DEBUG
        if(lane == 0) std::fprintf(stderr, "band rectangle: walk ended at (%d, %d): wx %d wy %d shift %d band [%d, %d] endFixed %d\n", i, j, wx, wy, shift, bandMin, bandMax, int(endFixed));
#endif
        return -2;
    }
    return found;
}

template<int TIE, bool BIG>
__global__ void __launch_bounds__(64)
sparseAnchorKernel(const uint32_t* __restrict__ kmerIds, const PairDesc* __restrict__ pairs, const DpTask* __restrict__ tasks,
    const uint32_t* __restrict__ ambiguousList, DpControl* __restrict__ control,
    const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ linkWords, const uint32_t* __restrict__ inBand, uint8_t* __restrict__ state,
    const uint32_t* __restrict__ hitMeta, const DpEnd* __restrict__ ends,
    const uint64_t* __restrict__ ordOffsets, uint32_t* __restrict__ ordScratch, DpResult* __restrict__ results, uint32_t* __restrict__ bigList, uint32_t listStride)
{
    using Shared = std::conditional_t<BIG, AnchorSharedBig, AnchorShared>;
    extern __shared__ uint32_t ldsWords[];
    Shared& sh = *reinterpret_cast<Shared*>(ldsWords);
    const int lane = laneId();
    // (the second launch: the tasks the first one listed because a rectangle did not fit it -- behind the first list, from `listStride` on)
    const uint32_t count = BIG ? control->anchorBigCount : control->ambiguousCount;
    if(BIG) ambiguousList += listStride;
    unsigned long long walked = 0;
    for(uint32_t slot = blockIdx.x; slot < count; slot += gridDim.x) {
        const uint32_t t = ambiguousList[slot];
        const DpTask task = tasks[t];
        const PairDesc pd = pairs[task.pair];
        const bool swapped = (hitMeta[task.pair] >> 31) != 0;
        const int32_t n = int32_t(inBand[t]);
        const int32_t lo = swapped ? task.bandMin : -task.bandMax;
        const uint32_t* __restrict__ const list = sorted + sparseListBase(ordOffsets, t);
        const uint32_t* __restrict__ const links = linkWords + sparseListBase(ordOffsets, t);
        const int32_t firstEnd = ends[t].bestI;
        const int32_t chunks = (n + WAVE - 1) / WAVE;
        walked += uint32_t(n);
        auto giveUp = [&](int why) { if(lane == 0) { state[t] = SPARSE_DENSE; noteGiveUp(control, GIVE_UP_FAR_LINK + (why == 4 ? 0 : why), pd, task); ANCHOR_REASON(why, n); } };
        if(chunks > ANCHOR_MAX_BITWORDS) { giveUp(3); continue; }           // (more hits than the bit words hold: only the lane kernel lists that many)
        waveLdsSync();                                       // (the task before has left the shared arrays)
        // ---- sweep: live hits and anchors ----
        uint32_t window = 0;                                 // bit d: the hit d before the current one has a link from a live later hit
        bool entered = false;                                // a live later hit begins an optimal chain
        int32_t anchorCount = 0;
        bool farLink = false;                                // a live hit with an optimal link further back than its link word tells
        uint32_t ahead = (chunks - 1) * WAVE + lane < n ? links[(chunks - 1) * WAVE + lane] : 0u;       // (the chunk after this one is on its way while this one is walked)
        for(int32_t c = chunks - 1; c >= 0; c--) {
            const uint32_t word = ahead;
            ahead = (c > 0 && (c - 1) * WAVE + lane < n) ? links[(c - 1) * WAVE + lane] : 0u;
            uint64_t liveBits = 0, anchorBits = 0;
            // Nearly every hit's only optimal link is the hit before it (link word 2): while the sweep is in its steady state -- the
            // hit it reaches is live through the link of the one behind it, nothing else is pending, no chain has begun at the border,
            // the first optimal end is not before it -- such a hit is live, an anchor, and leaves the state as it was: whole runs of
            // them are bit ranges of one ballot, and the scalar steps below are left with the hits that have something else to say
            // (about 15 of a task's 700; until round 5 every hit took its ~30 scalar instructions: 7.7e8 per launch).
            const int chunkHits = min(WAVE, n - c * WAVE);
            const uint64_t plain = ballot64(lane < chunkHits && (word & 0x7fffffffu) == 2u);
            for(int a = chunkHits - 1; a >= 0; a--) {
                if(window == 1u && !entered && c * WAVE + a <= firstEnd) {
                    const uint64_t others = ~plain & (a >= 63 ? ~0ULL : ((2ULL << a) - 1ULL));      // lanes up to a that are not plain
                    const int runLow = others ? 64 - __clzll((unsigned long long)others) : 0;         // the run of plain lanes that ends at a begins here
                    if(runLow <= a) {
                        const uint64_t run = (a >= 63 ? ~0ULL : ((2ULL << a) - 1ULL)) & ~(runLow ? ((1ULL << runLow) - 1ULL) : 0ULL);
                        liveBits |= run; anchorBits |= run; anchorCount += a - runLow + 1;
                        a = runLow;                       // (the loop's own step takes it below the run)
                        continue;
                    }
                }
                const uint32_t w = __builtin_amdgcn_readlane(word, a);
                const int32_t k = c * WAVE + a;
                const bool isLive = (window & 1u) != 0 || ((w >> 31) != 0 && k >= firstEnd);
                if(isLive) {
                    liveBits |= 1ULL << a;
                    if((window >> 1) == 0 && !entered && k <= firstEnd) { anchorBits |= 1ULL << a; ++anchorCount; }
                    window |= w & 0x3ffffffeu;
                    farLink = farLink || (w & 0x40000000u) != 0;
                    entered = entered || (w & 1u) != 0;
                }
                window >>= 1;
            }
            if(lane == 0) { sh.live[c] = liveBits; sh.anchor[c] = anchorBits; }
        }
        waveLdsSync();
        if(farLink) { giveUp(4); continue; }
        if(anchorCount == 0) { giveUp(1); continue; }
        // ---- windows: runs of live hits that are not anchors, from the last one back ----
        int32_t windows = 0;
        bool tooMany = false;
        {
            int32_t later = n;                               // the anchor met last; n: the border
            bool dirty = false;
            for(int32_t c = chunks - 1; c >= 0; c--) {
                uint64_t a = sh.anchor[c], o = sh.live[c] & ~a;
                while(a | o) {
                    if(dirty) {
                        if(a == 0) break;
                        const int top = 63 - __clzll((unsigned long long)a);
                        if(windows < ANCHOR_MAX_WINDOWS) { if(lane == 0) { sh.windowFrom[windows] = c * WAVE + top; sh.windowTo[windows] = later; } }
                        else tooMany = true;
                        ++windows;
                        later = c * WAVE + top; dirty = false;
                        a &= ~bitsFrom(top); o &= ~bitsFrom(top);
                    } else {
                        if(o == 0) { later = c * WAVE + (__ffsll((unsigned long long)a) - 1); break; }
                        const int top = 63 - __clzll((unsigned long long)o);
                        const uint64_t above = a & bitsFrom(top + 1);
                        if(above) later = c * WAVE + (__ffsll((unsigned long long)above) - 1);
                        dirty = true;
                        a &= ~bitsFrom(top); o &= ~bitsFrom(top);
                    }
                }
            }
            if(dirty) {
                if(windows < ANCHOR_MAX_WINDOWS) { if(lane == 0) { sh.windowFrom[windows] = -1; sh.windowTo[windows] = later; } }
                else tooMany = true;
                ++windows;
            }
        }
        if(tooMany) { giveUp(2); continue; }
        waveLdsSync();
        // ---- the rectangles ----
        auto hitAt = [&](int32_t k, int32_t& x, int32_t& y) {
            const uint32_t e = list[k];
            const int32_t hp = int32_t(e >> 17), hs = hp + lo + int32_t((e >> 7) & 1023u);
            x = swapped ? hs : hp; y = swapped ? hp : hs;
        };
        const uint32_t* __restrict__ const p0 = kmerIds + pd.begin0;
        const uint32_t* __restrict__ const p1 = kmerIds + pd.begin1;
        uint32_t windowPairsTotal = 0;
        bool fits = true, tooBig = false;
        int why = GIVE_UP_RECTANGLE;
        for(int32_t w = 0; w < windows && fits; w++) {
            const int32_t from = sh.windowFrom[w], to = sh.windowTo[w];
            int32_t x0 = 0, y0 = 0, x1 = int32_t(pd.nx) - 1, y1 = int32_t(pd.ny) - 1;
            if(from >= 0) { hitAt(from, x0, y0); ++x0; ++y0; }
            if(to < n) { hitAt(to, x1, y1); --x1; --y1; }
            const int32_t wx = x1 - x0 + 1, wy = y1 - y0 + 1;
            if(wx < 1 || wy < 1 || wx > Shared::maxSide || wy > Shared::maxSide || (!Shared::bandTrace && (wx + 1) * (wy + 1) > Shared::maxCells)) { fits = false; tooBig = true; break; }
            int32_t found;
            if constexpr (Shared::bandTrace) found = anchorRectangleBand<TIE>(sh, p0, p1, x0, x1, y0, y1, from >= 0, to < n, task.bandMin, task.bandMax, windowPairsTotal, lane);
            else found = anchorRectangle<TIE>(sh, p0, p1, x0, x1, y0, y1, from >= 0, to < n, task.bandMin, task.bandMax, windowPairsTotal, lane);
            if(found < 0) { fits = false; tooBig = true; why = found == -1 ? GIVE_UP_RECTANGLE_PAIRS : (found == -2 ? GIVE_UP_RECTANGLE_WALK : GIVE_UP_RECTANGLE); break; }       // (the pairs did not fit, or the walk contradicted the anchors: the larger form tries once more)
            if(lane == 0) { sh.windowBegin[w] = windowPairsTotal; sh.windowPairs[w] = uint32_t(found); }
            windowPairsTotal += uint32_t(found);
        }
        if(!BIG && tooBig && bigList) {
            // (stays SPARSE_AMBIGUOUS: the second launch's)
            if(lane == 0) bigList[listStride + atomicAdd(&control->anchorBigCount, 1u)] = t;
            continue;
        }
        if(!fits || uint32_t(anchorCount) + windowPairsTotal > min(pd.nx, pd.ny)) {
            if(lane == 0) { state[t] = SPARSE_DENSE; noteGiveUp(control, fits ? int(GIVE_UP_PAIR_TOTAL) : why, pd, task); ANCHOR_REASON(3, n); }
            continue;
        }
        waveLdsSync();
        // ---- emit: from the end of the task's range downwards ----
        const uint64_t ordBase = ordOffsets[t];
        uint32_t pos = min(pd.nx, pd.ny);
        auto emitWindow = [&](int32_t w) {
            const uint32_t begin = sh.windowBegin[w], found = sh.windowPairs[w];
            for(uint32_t a = uint32_t(lane); a < found; a += WAVE) {
                const uint32_t e = sh.pairs[begin + a];
                *reinterpret_cast<uint2*>(ordScratch + 2 * (ordBase + pos - 1u - a)) = make_uint2(e >> 16, e & 0xffffu);
            }
            pos -= found;
        };
        int32_t next = 0;                                    // the windows are in the order they are met
        if(next < windows && sh.windowTo[next] >= n) { emitWindow(next); ++next; }
        uint32_t entryAhead = (chunks - 1) * WAVE + lane < n ? list[(chunks - 1) * WAVE + lane] : 0u;     // (as in the sweep: the next chunk is on its way)
        for(int32_t c = chunks - 1; c >= 0; c--) {
            const uint32_t entry = entryAhead;
            entryAhead = (c > 0 && (c - 1) * WAVE + lane < n) ? list[(c - 1) * WAVE + lane] : 0u;
            const int32_t hp = int32_t(entry >> 17), hs = hp + lo + int32_t((entry >> 7) & 1023u);
            const int32_t x = swapped ? hs : hp, y = swapped ? hp : hs;
            uint64_t todo = sh.anchor[c];
            auto emitAnchors = [&](uint64_t part) {
                if((part >> lane) & 1ULL) {
                    const uint32_t rank = uint32_t(__popcll(part & bitsFrom(lane + 1)));
                    *reinterpret_cast<uint2*>(ordScratch + 2 * (ordBase + pos - 1u - rank)) = make_uint2(uint32_t(x), uint32_t(y));
                }
                pos -= uint32_t(__popcll(part));
            };
            while(next < windows && sh.windowTo[next] >= c * WAVE) {
                // The window lies before the anchor windowTo: that anchor and those behind it first, then the window's pairs.
                const uint64_t part = todo & bitsFrom(sh.windowTo[next] - c * WAVE);
                emitAnchors(part); todo &= ~part;
                emitWindow(next); ++next;
            }
            emitAnchors(todo);
        }
        if(lane == 0) {
            DpEnd e; e.traceOffset = 0; e.bestI = e.bestJ = 0; e.score = ends[t].score; e.laneBase = 0; e.bundleIterations = 0; e.pad = 0;
            tracebackFinish(pos, pd, e, ordBase, t, results);
            state[t] = SPARSE_CERTIFIED;
        }
    }
    if(lane == 0 && walked) atomicAdd(&control->ambiguousHits, walked);
}
