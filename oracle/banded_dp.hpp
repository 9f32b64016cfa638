// TEST INFRASTRUCTURE ONLY -- nothing under shasta_amd/ may include, link or call this.
//
// CPU restatement of the one piece of the Align4 path that is NOT Shasta code:
//
//   seqan::globalAlignment(graph, Score<int,Simple>(match, mismatch, gap),
//                          AlignConfig<true,true,true,true>(), bandMin, bandMax, LinearGaps())
//
// as called at /root/reference/src/Align4.cpp:1028-1033 (SeqAn 2.4.0, Ubuntu
// package libseqan2-dev; docs/Prerequisites.html:110).  SeqAn is an un-vendored
// third-party dependency and is absent from this container, so this restates
// its published algorithm (banded Needleman-Wunsch overlap alignment, linear
// gaps, all four end-gap classes free) and anchors parity on the reference's
// call site and on its post-processing loop (src/Align4.cpp:1041-1068).
//
// *** PARITY UNPINNED for tie handling. ***  The recurrence and the band
// geometry are unambiguous.  What SeqAn decides internally and no reference
// test pins is (a) the direction kept when two predecessors tie and (b) the
// end cell kept when several last-row/last-column cells tie on the maximum.
// The policy below is the author's reading of SeqAn 2.4.0
// (dp_formula_linear.h: diagonal, then vertical, then horizontal on strict '<';
//  dp_scout.h: first strict '>' in column-major scan order).
// It is written as data (TiePolicy) so that it can be changed in one place if
// a true SeqAn run ever disagrees.  The HIP kernel implements the same policy.
//
// A second reading, made independently in round 3 (again from memory of the 2.4.0
// sources, which are not in this container), agrees with it and names the code:
//   * globalAlignment(Graph<Alignment<..>>&, Score, AlignConfig, lowerDiag, upperDiag, LinearGaps) sets up
//     AlignConfig2<DPGlobal, DPBandConfig<BandOn>, FreeEndGaps_<..>, TracebackOn<TracebackConfig_<SingleTrace, GapsLeft>>>:
//     ONE trace bit per cell, so the order in which _computeTraceback tests the bits cannot matter;
//   * dp_formula_linear.h, _computeScore(.., RecursionDirectionAll, ..): the vertical and the horizontal candidate meet in
//     _maxScore(target, vertical, horizontal, ..) -- "leftCompare < rightCompare ? right : left", the left (vertical) keeps a
//     tie -- and the winner meets the diagonal in _maxScore(target, diagonal, gap, DIAGONAL, tv): the left (diagonal) keeps a
//     tie.  Hence diagonal >= vertical >= horizontal.  On the band's edge cells (RecursionDirectionUpperDiagonal /
//     LowerDiagonal) the missing neighbour simply does not take part, which is what -infinity outside the band gives here;
//   * dp_scout.h, _scoutBestScore: "if(_scoreOfCell(activeCell) > _scoreOfCell(dpScout._maxScore))" -- strictly greater --
//     over the tracked cells (last row and last column: all four end gaps free) in the order the cells are computed,
//     column by column of the horizontal sequence, rows ascending: the first maximum in (i, j) order.
// It remains a reading: oracle/census.py counts what depends on it.
//
// Geometry.  seq0 (nx markers) is SeqAn's horizontal sequence, seq1 (ny) the
// vertical one.  DP cell (i,j) = i symbols of seq0 and j of seq1 consumed,
// 0<=i<=nx, 0<=j<=ny.  A cell is inside the band iff bandMin <= i-j <= bandMax
// (a diagonal step into (i,j) aligns x=i-1 with y=j-1 and x-y = i-j, which is
// how src/Align4.cpp:908-917 derives the band from Y = nx-1+y-x).
// Free end gaps: row 0 and column 0 are 0; the result is the maximum over the
// in-band cells of the last row (j=ny) and last column (i=nx).
#ifndef ORACLE_BANDED_DP_HPP
#define ORACLE_BANDED_DP_HPP

#include <cstdint>
#include <cstdlib>
#include <limits>
#include <utility>
#include <vector>

namespace oracle {

struct TiePolicy {
    // Direction priority when scores tie: lower number wins.
    // SeqAn 2.4.0 reading: diagonal(0) < vertical(1) < horizontal(2).
    int diagonalRank = 0;
    int verticalRank = 1;
    int horizontalRank = 2;
    // End cell: scan columns i=0..nx, inside a column rows j ascending; a later
    // cell replaces the current best only if strictly greater.
    bool firstMaximumWins = true;
};

// The policy in force in this process (each shared library that includes this header has its own copy and its own
// setter: oracle_set_tie_policy / ref_set_tie_policy).  Default-constructed = the reading above.  The tie census of
// bench.py and tests/test_tie_census.py run the aligner under every other policy to count what depends on the reading.
// Policies by number: index = 2 * order + (lastMaximumWins ? 1 : 0), order over the six priority orders of
// (diagonal, vertical, horizontal): 0 DVH (the reading), 1 DHV, 2 VDH, 3 VHD, 4 HDV, 5 HVD.
inline TiePolicy tiePolicyByIndex(int index)
{
    static const int ranks[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {2, 0, 1}, {1, 2, 0}, {2, 1, 0}};   // {diagonal, vertical, horizontal} rank
    TiePolicy p;
    const int order = (index / 2) % 6;
    p.diagonalRank = ranks[order][0]; p.verticalRank = ranks[order][1]; p.horizontalRank = ranks[order][2];
    p.firstMaximumWins = (index & 1) == 0;
    return p;
}
// (ORACLE_TIE_POLICY=<n> in the environment: the policy a process starts with -- tests/test_seqan_pin_kit.py runs the pin kit's
// program against the SeqAn stand-in under several policies that way.)
inline TiePolicy& activeTiePolicy()
{
    static TiePolicy policy = [] { const char* e = std::getenv("ORACLE_TIE_POLICY"); return (e && *e) ? tiePolicyByIndex(std::atoi(e)) : TiePolicy(); }();
    return policy;
}

enum TraceOp : uint8_t { TRACE_NONE = 0, TRACE_DIAG = 1, TRACE_VERT = 2, TRACE_HORI = 3 };

struct BandedDpResult {
    bool ok = false;              // false <=> SeqAn would return MinValue<int>::VALUE
    int32_t score = std::numeric_limits<int32_t>::min();
    uint32_t iEnd = 0, jEnd = 0;  // end cell of the traceback (max cell)
    uint32_t iBegin = 0, jBegin = 0; // cell where the traceback stopped (row 0 or column 0)
    // Operations from (iBegin,jBegin) to (iEnd,jEnd), in forward order.
    std::vector<uint8_t> ops;
};

// Vertical step consumes a symbol of seq1 (gap in row 0), horizontal a symbol of seq0.
template<class T>
inline void bandedOverlapAlignment(
    const T* seq0, uint32_t nx,
    const T* seq1, uint32_t ny,
    int32_t matchScore, int32_t mismatchScore, int32_t gapScore,
    int32_t bandMin, int32_t bandMax,
    BandedDpResult& result,
    const TiePolicy& policy = activeTiePolicy())
{
    result = BandedDpResult();
    if(bandMin > bandMax) return;
    // The band must intersect the matrix: diagonals range over [-ny, nx].
    if(bandMin > int32_t(nx) || bandMax < -int32_t(ny)) return;

    const int64_t W = int64_t(bandMax) - int64_t(bandMin) + 1;   // band width in diagonals
    const int32_t NEG = std::numeric_limits<int32_t>::min() / 4;

    // trace[i*W + b], b = (i-j) - bandMin.
    std::vector<uint8_t> trace(size_t(nx + 1) * size_t(W), TRACE_NONE);
    std::vector<int32_t> prev(size_t(W), NEG), cur(size_t(W), NEG);

    bool haveBest = false;
    int32_t best = 0;
    uint32_t bestI = 0, bestJ = 0;
    auto scout = [&](int32_t s, uint32_t i, uint32_t j) {
        if(!haveBest || s > best) { haveBest = true; best = s; bestI = i; bestJ = j; }
    };

    for(uint32_t i = 0; i <= nx; i++) {
        // Rows of column i inside the band: i-bandMax <= j <= i-bandMin.
        const int64_t jLo64 = std::max<int64_t>(0, int64_t(i) - bandMax);
        const int64_t jHi64 = std::min<int64_t>(ny, int64_t(i) - bandMin);
        std::fill(cur.begin(), cur.end(), NEG);
        if(jLo64 <= jHi64) {
            for(int64_t j64 = jLo64; j64 <= jHi64; j64++) {
                const uint32_t j = uint32_t(j64);
                const int64_t b = (int64_t(i) - j64) - bandMin;
                int32_t s;
                uint8_t t = TRACE_NONE;
                if(i == 0 || j == 0) {
                    s = 0;                       // free leading gaps
                } else {
                    const int32_t dScore = (prev[size_t(b)] == NEG) ? NEG :
                        prev[size_t(b)] + ((seq0[i-1] == seq1[j-1]) ? matchScore : mismatchScore);
                    // horizontal: from (i-1,j), diagonal index b-1 of previous column.
                    const int32_t hScore = (b - 1 >= 0 && prev[size_t(b-1)] != NEG) ?
                        prev[size_t(b-1)] + gapScore : NEG;
                    // vertical: from (i,j-1), diagonal index b+1 of this column.
                    const int32_t vScore = (b + 1 < W && cur[size_t(b+1)] != NEG) ?
                        cur[size_t(b+1)] + gapScore : NEG;
                    // Pick by score, ties by rank.
                    s = dScore; t = TRACE_DIAG; int r = policy.diagonalRank;
                    if(vScore > s || (vScore == s && policy.verticalRank < r)) {
                        s = vScore; t = TRACE_VERT; r = policy.verticalRank;
                    }
                    if(hScore > s || (hScore == s && policy.horizontalRank < r)) {
                        s = hScore; t = TRACE_HORI; r = policy.horizontalRank;
                    }
                    if(s <= NEG) { s = NEG; t = TRACE_NONE; }   // unreachable cell
                }
                cur[size_t(b)] = s;
                trace[size_t(i) * size_t(W) + size_t(b)] = t;
                if(s != NEG && (j == ny || i == nx)) {
                    if(policy.firstMaximumWins) scout(s, i, j);
                    else if(!haveBest || s >= best) { haveBest = true; best = s; bestI = i; bestJ = j; }
                }
            }
        }
        prev.swap(cur);
    }

    if(!haveBest) return;
    result.ok = true;
    result.score = best;
    result.iEnd = bestI;
    result.jEnd = bestJ;

    // Traceback.
    uint32_t i = bestI, j = bestJ;
    std::vector<uint8_t> reversed;
    while(i > 0 && j > 0) {
        const int64_t b = (int64_t(i) - int64_t(j)) - bandMin;
        const uint8_t t = trace[size_t(i) * size_t(W) + size_t(b)];
        if(t == TRACE_DIAG) { --i; --j; }
        else if(t == TRACE_VERT) { --j; }
        else if(t == TRACE_HORI) { --i; }
        else break;
        reversed.push_back(t);
    }
    result.iBegin = i;
    result.jBegin = j;
    result.ops.assign(reversed.rbegin(), reversed.rend());
}

// The post-processing of src/Align4.cpp:1051-1068 expressed on the operation
// list: a column contributes (ordinal0, ordinal1) iff it is a diagonal step and
// the two kmer ids are equal.  (Leading/trailing free gaps contain no diagonal
// step, and the reference loop's early exit only cuts trailing gap columns.)
template<class T>
inline void diagonalMatches(
    const T* seq0, const T* seq1, const BandedDpResult& r,
    std::vector< std::pair<uint32_t, uint32_t> >& ordinals)
{
    ordinals.clear();
    if(!r.ok) return;
    uint32_t x = r.iBegin, y = r.jBegin;
    for(const uint8_t op : r.ops) {
        if(op == TRACE_DIAG) {
            if(seq0[x] == seq1[y]) ordinals.push_back(std::make_pair(x, y));
            ++x; ++y;
        } else if(op == TRACE_VERT) {
            ++y;
        } else {
            ++x;
        }
    }
}

}  // namespace oracle
#endif
