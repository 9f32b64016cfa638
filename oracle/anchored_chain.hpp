// TEST INFRASTRUCTURE ONLY -- nothing under shasta_amd/ may include, link or call this.
//
// Prototype of the step after the sparse path (oracle/sparse_chain.hpp, shasta_amd/csrc/align4_sparse.hpp): the banded alignment
// of a task whose optimal chain of matches is NOT unique, with the dense DP confined to the stretches where the chains differ.
// (The sparse path sends such a task to the dense DP as a whole; they are the long tasks, mostly, and a traceback launch is as
// long as its longest path.)  The device form is shasta_amd/csrc/align4_anchor.hpp; this file states the algorithm in the same steps.
//
//   forward   D(m) = 6 + value(m), value(m) = max(-min(x, y), max over hits m' before m in both ordinals of D(m') - chebyshev(m', m));
//             with every hit, WHICH terms attain the maximum: the border, and the set of predecessors (its optimal links)
//   ends      the hits with D(m) - min(nx - 1 - x, ny - 1 - y) = best
//   live      the hits on SOME optimal chain: an end, or an optimal predecessor of a live hit -- one sweep from the last hit back
//   anchors   the hits on EVERY optimal chain: live, and at the moment the sweep reaches the hit no link of a live later hit passes
//             over it (to an earlier hit, or to the border), and no end lies before it
//
// Between two consecutive anchors either nothing is live -- the chain steps from one anchor to the next, whatever the tie policy
// (a sub-chain that were the only one between two anchors would consist of anchors) -- or several sub-chains are: then the dense DP
// runs on the rectangle between the two anchors alone, under the policy, with its corners FIXED: it starts in the cell behind the
// first anchor (value 0, the two borders that leave it reached by gaps only) and the traceback starts in the cell in front of the
// second.  Before the first anchor and after the last the free border stays what it is.  Why the dense traceback decides inside the
// rectangle as it does inside the whole matrix: every cell of the traced path lies on an optimal path, all of which pass both
// anchors; a predecessor that ties for a cell's maximum is on an optimal path too, so inside the rectangle, and has the same value
// relative to the first anchor in both problems; one that does not tie has, in the rectangle, at most its value in the whole
// matrix (the rectangle's paths are paths of the whole matrix).  This file checks that claim: anchoredAlignment must equal
// bandedOverlapAlignment + diagonalMatches on EVERY task under EVERY policy (tests/test_oracle_anchored.py,
// scripts/sparse_census.py ... anchored).
#ifndef ORACLE_ANCHORED_CHAIN_HPP
#define ORACLE_ANCHORED_CHAIN_HPP

#include "banded_dp.hpp"
#include "sparse_chain.hpp"

#include <algorithm>
#include <cstdint>
#include <limits>
#include <unordered_map>
#include <utility>
#include <vector>

namespace oracle {

struct AnchoredResult {
    std::vector< std::pair<uint32_t, uint32_t> > ordinals;
    int32_t score = std::numeric_limits<int32_t>::min();
    uint64_t hits = 0, anchors = 0, windows = 0;
    uint64_t denseCells = 0;          // cells of the dense problems that were solved
    uint64_t largestWindow = 0;       // cells of the largest rectangle
    uint32_t longestLink = 0;         // how many hits back the furthest optimal link of a LIVE hit goes (the device's link word holds 29)
    bool wholeTaskDense = false;      // no anchor (or a tie with the empty alignment): the dense DP on the whole task, as today
};

namespace anchored_detail {
struct Hit { int32_t x, y; };
inline int32_t cheb(const Hit& a, const Hit& b) { return std::max(b.x - a.x - 1, b.y - a.y - 1); }
}

// The dense DP on markers [x0, x1] x [y0, y1] of the two reads inside the task's band; beginFixed: every path starts in the corner
// cell before (x0, y0); endFixed: the traceback starts in the corner cell behind (x1, y1).  The matches of the traced path, in order.
template<class T>
inline void cornerAlignment(const T* seq0, const T* seq1, int32_t x0, int32_t x1, int32_t y0, int32_t y1, bool beginFixed, bool endFixed,
    int32_t bandMin, int32_t bandMax, const TiePolicy& policy, std::vector< std::pair<uint32_t, uint32_t> >& out, uint64_t& cells)
{
    out.clear();
    const int32_t wx = x1 - x0 + 1, wy = y1 - y0 + 1;
    const int32_t NEG = std::numeric_limits<int32_t>::min() / 4;
    const size_t stride = size_t(wy) + 1;
    std::vector<int32_t> H((size_t(wx) + 1) * stride, NEG);
    std::vector<uint8_t> trace((size_t(wx) + 1) * stride, TRACE_NONE);
    auto inBand = [&](int32_t i, int32_t j) { const int32_t d = (i + x0) - (j + y0); return d >= bandMin && d <= bandMax; };
    bool haveBest = false; int32_t best = 0, bestI = 0, bestJ = 0;
    for(int32_t i = 0; i <= wx; i++) for(int32_t j = 0; j <= wy; j++) {
        if(!inBand(i, j)) continue;
        ++cells;
        int32_t s; uint8_t t = TRACE_NONE;
        if(i == 0 || j == 0) {
            if(!beginFixed) s = 0;
            else if(i == 0 && j == 0) s = 0;
            else if(i == 0) { s = H[size_t(j - 1)] == NEG ? NEG : H[size_t(j - 1)] - 1; t = TRACE_VERT; }
            else { s = H[size_t(i - 1) * stride] == NEG ? NEG : H[size_t(i - 1) * stride] - 1; t = TRACE_HORI; }
            if(s == NEG) t = TRACE_NONE;
        } else {
            const int32_t d0 = H[size_t(i - 1) * stride + size_t(j - 1)], h0 = H[size_t(i - 1) * stride + size_t(j)], v0 = H[size_t(i) * stride + size_t(j - 1)];
            const int32_t dScore = d0 == NEG ? NEG : d0 + (seq0[x0 + i - 1] == seq1[y0 + j - 1] ? 6 : -1);
            const int32_t hScore = h0 == NEG ? NEG : h0 - 1, vScore = v0 == NEG ? NEG : v0 - 1;
            s = dScore; t = TRACE_DIAG; int r = policy.diagonalRank;
            if(vScore > s || (vScore == s && policy.verticalRank < r)) { s = vScore; t = TRACE_VERT; r = policy.verticalRank; }
            if(hScore > s || (hScore == s && policy.horizontalRank < r)) { s = hScore; t = TRACE_HORI; r = policy.horizontalRank; }
            if(s <= NEG) { s = NEG; t = TRACE_NONE; }
        }
        H[size_t(i) * stride + size_t(j)] = s; trace[size_t(i) * stride + size_t(j)] = t;
        if(s != NEG && (i == wx || j == wy)) {
            if(!haveBest || s > best || (s == best && !policy.firstMaximumWins)) { haveBest = true; best = s; bestI = i; bestJ = j; }
        }
    }
    int32_t i = endFixed ? wx : bestI, j = endFixed ? wy : bestJ;
    if(!endFixed && !haveBest) return;
    while(i > 0 || j > 0) {
        if(!beginFixed && (i == 0 || j == 0)) break;
        const uint8_t t = trace[size_t(i) * stride + size_t(j)];
        if(t == TRACE_DIAG) { --i; --j; if(seq0[x0 + i] == seq1[y0 + j]) out.push_back(std::make_pair(uint32_t(x0 + i), uint32_t(y0 + j))); }
        else if(t == TRACE_VERT) --j;
        else if(t == TRACE_HORI) --i;
        else break;
    }
    std::reverse(out.begin(), out.end());
}

template<class T>
inline void anchoredAlignment(const T* seq0, uint32_t nx, const T* seq1, uint32_t ny, int32_t bandMin, int32_t bandMax,
    AnchoredResult& r, const TiePolicy& policy = activeTiePolicy())
{
    using anchored_detail::Hit;
    using anchored_detail::cheb;
    r = AnchoredResult();
    auto wholeTask = [&]() {
        BandedDpResult dp;
        bandedOverlapAlignment(seq0, nx, seq1, ny, 6, -1, -1, bandMin, bandMax, dp, policy);
        diagonalMatches(seq0, seq1, dp, r.ordinals);
        r.score = dp.ok ? dp.score : std::numeric_limits<int32_t>::min();
        r.wholeTaskDense = true;
        r.denseCells += uint64_t(nx) * uint64_t(int64_t(bandMax) - bandMin + 1);
    };
    if(bandMin > bandMax || bandMin > int32_t(nx) || bandMax < -int32_t(ny)) { wholeTask(); return; }
    std::vector<Hit> hits;
    {
        std::unordered_map<T, std::vector<uint32_t> > where;
        for(uint32_t y = 0; y < ny; y++) where[seq1[y]].push_back(y);
        for(uint32_t x = 0; x < nx; x++) {
            const auto it = where.find(seq0[x]);
            if(it == where.end()) continue;
            for(const uint32_t y : it->second) {
                const int64_t d = int64_t(x) - int64_t(y);
                if(d >= bandMin && d <= bandMax) hits.push_back(Hit{int32_t(x), int32_t(y)});
            }
        }
    }
    const int32_t n = int32_t(hits.size());
    const size_t count = static_cast<size_t>(n);
    r.hits = uint64_t(n);
    const int32_t Z = bestMatchlessScore(nx, ny, bandMin, bandMax);
    if(n == 0) { r.score = Z; return; }
    const int32_t NEG = std::numeric_limits<int32_t>::min() / 4;
    // Forward, with every hit's optimal links.
    std::vector<int32_t> D(count), prefixMax(count);
    std::vector< std::vector<int32_t> > links(count);
    std::vector<char> fromBorder(count, 0), endCandidate(count, 0);
    int32_t best = NEG, firstEnd = -1;
    for(int32_t k = 0; k < n; k++) {
        const Hit m = hits[size_t(k)];
        int32_t v = -std::min(m.x, m.y);
        bool border = true;
        std::vector<int32_t>& mine = links[size_t(k)];
        for(int32_t q = k - 1; q >= 0; q--) {
            if(prefixMax[size_t(q)] - (m.x - hits[size_t(q)].x - 1) < v) break;
            if(hits[size_t(q)].x < m.x && hits[size_t(q)].y < m.y) {
                const int32_t candidate = D[size_t(q)] - cheb(hits[size_t(q)], m);
                if(candidate > v) { v = candidate; border = false; mine.clear(); mine.push_back(q); }
                else if(candidate == v) mine.push_back(q);
            }
        }
        D[size_t(k)] = 6 + v; fromBorder[size_t(k)] = border;
        prefixMax[size_t(k)] = k ? std::max(prefixMax[size_t(k) - 1], D[size_t(k)]) : D[size_t(k)];
        const int32_t end = D[size_t(k)] - std::min(int32_t(nx) - 1 - m.x, int32_t(ny) - 1 - m.y);
        if(end > best) { best = end; firstEnd = k; }
        endCandidate[size_t(k)] = end >= best;          // an end if no later hit improves on `best`: those from firstEnd on
    }
    if(best < Z) { r.score = Z; return; }                       // every optimal path is without a match
    if(best == Z) { wholeTask(); return; }                      // a chain ties with the empty alignment: the policy decides
    r.score = best;
    // Live hits and anchors: one sweep from the last hit back.  pending[q] > 0: a live hit met so far links to q.
    std::vector<char> live(count, 0), anchor(count, 0);
    std::vector<char> pending(count, 0);
    int32_t open = 0;                 // hits before the current one that a live later hit links to
    bool enteredLater = false;        // a live later hit is the first of an optimal chain
    for(int32_t k = n - 1; k >= 0; k--) {
        if(pending[size_t(k)]) --open;
        const bool isLive = pending[size_t(k)] || (endCandidate[size_t(k)] && k >= firstEnd);
        if(!isLive) continue;
        live[size_t(k)] = 1;
        anchor[size_t(k)] = open == 0 && !enteredLater && k <= firstEnd;
        for(const int32_t q : links[size_t(k)]) { r.longestLink = std::max(r.longestLink, uint32_t(k - q)); if(!pending[size_t(q)]) { pending[size_t(q)] = 1; ++open; } }
        enteredLater = enteredLater || fromBorder[size_t(k)];
    }
    // The chain: anchors as they are, a rectangle wherever something else is live between two of them.
    std::vector< std::pair<uint32_t, uint32_t> > reversed, piece;
    int32_t later = -1;               // the anchor met last (the next one in the chain); -1: the border
    bool dirty = false;
    bool any = false;
    auto rectangle = [&](int32_t from, int32_t to) {
        const int32_t x0 = from < 0 ? 0 : hits[size_t(from)].x + 1, y0 = from < 0 ? 0 : hits[size_t(from)].y + 1;
        const int32_t x1 = to < 0 ? int32_t(nx) - 1 : hits[size_t(to)].x - 1, y1 = to < 0 ? int32_t(ny) - 1 : hits[size_t(to)].y - 1;
        uint64_t cells = 0;
        cornerAlignment(seq0, seq1, x0, x1, y0, y1, from >= 0, to >= 0, bandMin, bandMax, policy, piece, cells);
        r.denseCells += cells; r.windows += 1; r.largestWindow = std::max(r.largestWindow, uint64_t(x1 - x0 + 2) * uint64_t(y1 - y0 + 2));
        for(size_t a = piece.size(); a-- > 0;) reversed.push_back(piece[a]);
    };
    for(int32_t k = n - 1; k >= 0; k--) {
        if(!live[size_t(k)]) continue;
        if(!anchor[size_t(k)]) { dirty = true; continue; }
        if(dirty) rectangle(k, later);
        reversed.push_back(std::make_pair(uint32_t(hits[size_t(k)].x), uint32_t(hits[size_t(k)].y)));
        later = k; dirty = false; any = true; r.anchors += 1;
    }
    if(!any) { r.anchors = 0; r.windows = 0; r.denseCells = 0; wholeTask(); return; }
    if(dirty) rectangle(-1, later);
    r.ordinals.assign(reversed.rbegin(), reversed.rend());
}

}  // namespace oracle
#endif
