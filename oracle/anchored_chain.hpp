// TEST INFRASTRUCTURE ONLY -- nothing under shasta_amd/ may include, link or call this.
//
// Prototype of the NEXT step after the sparse path (oracle/sparse_chain.hpp, shasta_amd/csrc/align4_sparse.hpp): the banded
// alignment of a task whose optimal chain of matches is NOT unique, with the dense DP confined to the stretches where the chains
// differ.  (The sparse path sends such a task to the dense DP as a whole; they are the long tasks, mostly, and a traceback launch is
// as long as its longest path.)
//
//   forward   D(m): best score of a chain that ends with hit m              (sparse_chain.hpp's recurrence)
//   backward  T(m): best score of what can follow m (further hits, then the way out to the free border)
//   O         the hits on SOME optimal chain: D(m) + T(m) = best
//   links     m' -> m optimal: both in O and D(m') - chebyshev(m', m) attains m's maximum; border -> m; m -> border
//   anchors   hits on EVERY optimal chain: in O, alone in O at their ordinal x, and no optimal link passes over x
//
// Between two consecutive anchors (and between the border and the first / the last anchor and the border) either ONE optimal
// sub-chain exists -- then it is the answer there, whatever the tie policy -- or several: then the dense DP runs on that stretch
// alone, under the policy, with its corners FIXED at the anchors by padding both sub-sequences with a run of synthetic markers that
// match pairwise (so long that no path gains by leaving it).  Why the dense traceback decides inside the padded stretch as it does
// inside the whole matrix: every cell of the traced path lies on an optimal path, all of which pass the anchors; a predecessor
// that ties for a cell's maximum is on an optimal path too (through the anchors: same value relative to the anchor in both
// problems); one that does not tie has, in the padded problem, at most the value of a path through the anchor (unchanged) or of a
// path that enters from the free border and forgoes the padding (far below).  This file checks that claim: anchoredAlignment must
// equal bandedOverlapAlignment + diagonalMatches on EVERY task under EVERY policy (tests/test_oracle_golden.py,
// scripts/sparse_census.py --anchored).
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
    uint64_t denseCells = 0;          // cells of the dense problems that were solved (padding included)
    bool wholeTaskDense = false;      // no anchor (or a tie with the empty alignment): the dense DP on the whole task, as today
};

namespace anchored_detail {
struct Hit { int32_t x, y; };
inline int32_t cheb(const Hit& a, const Hit& b) { return std::max(b.x - a.x - 1, b.y - a.y - 1); }
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
    r.hits = uint64_t(n);
    const int32_t Z = bestMatchlessScore(nx, ny, bandMin, bandMax);
    if(n == 0) { r.score = Z; return; }
    const int32_t NEG = std::numeric_limits<int32_t>::min() / 4;
    auto enter = [&](const Hit& m) { return -std::min(m.x, m.y); };
    auto leave = [&](const Hit& m) { return -std::min(int32_t(nx) - 1 - m.x, int32_t(ny) - 1 - m.y); };
    // Forward.
    const size_t count = static_cast<size_t>(n);
    std::vector<int32_t> value(count), D(count), prefixMax(count);
    for(int32_t k = 0; k < n; k++) {
        int32_t v = enter(hits[size_t(k)]);
        for(int32_t q = k - 1; q >= 0; q--) {
            if(prefixMax[size_t(q)] - (hits[size_t(k)].x - hits[size_t(q)].x - 1) < v) break;
            if(hits[size_t(q)].x < hits[size_t(k)].x && hits[size_t(q)].y < hits[size_t(k)].y) v = std::max(v, D[size_t(q)] - cheb(hits[size_t(q)], hits[size_t(k)]));
        }
        value[size_t(k)] = v; D[size_t(k)] = 6 + v;
        prefixMax[size_t(k)] = k ? std::max(prefixMax[size_t(k) - 1], D[size_t(k)]) : D[size_t(k)];
    }
    // Backward: T(m) = max(leave(m), max over later hits m'' of 6 + T(m'') - cheb(m, m'')).
    std::vector<int32_t> Tm(count), suffixMax(count);
    for(int32_t k = n - 1; k >= 0; k--) {
        int32_t t = leave(hits[size_t(k)]);
        for(int32_t q = k + 1; q < n; q++) {
            if(suffixMax[size_t(q)] - (hits[size_t(q)].x - hits[size_t(k)].x - 1) < t) break;
            if(hits[size_t(q)].x > hits[size_t(k)].x && hits[size_t(q)].y > hits[size_t(k)].y) t = std::max(t, 6 + Tm[size_t(q)] - cheb(hits[size_t(k)], hits[size_t(q)]));
        }
        Tm[size_t(k)] = t;
        suffixMax[size_t(k)] = (k + 1 < n) ? std::max(suffixMax[size_t(k) + 1], 6 + t) : 6 + t;
    }
    int32_t best = NEG;
    for(int32_t k = 0; k < n; k++) best = std::max(best, D[size_t(k)] + Tm[size_t(k)]);
    if(best < Z) { r.score = Z; return; }                       // every optimal path is without a match
    if(best == Z) { wholeTask(); return; }                      // a chain ties with the empty alignment: the policy decides
    r.score = best;
    // The hits on some optimal chain, their optimal links.
    std::vector<char> optimal(count, 0);
    for(int32_t k = 0; k < n; k++) optimal[size_t(k)] = D[size_t(k)] + Tm[size_t(k)] == best;
    std::vector< std::vector<int32_t> > predecessors(count);
    std::vector<char> fromBorder(count, 0), toBorder(count, 0);
    for(int32_t k = 0; k < n; k++) {
        if(!optimal[size_t(k)]) continue;
        fromBorder[size_t(k)] = enter(hits[size_t(k)]) == value[size_t(k)];
        toBorder[size_t(k)] = leave(hits[size_t(k)]) == Tm[size_t(k)];
        for(int32_t q = k - 1; q >= 0; q--) {
            if(prefixMax[size_t(q)] - (hits[size_t(k)].x - hits[size_t(q)].x - 1) < value[size_t(k)]) break;
            if(optimal[size_t(q)] && hits[size_t(q)].x < hits[size_t(k)].x && hits[size_t(q)].y < hits[size_t(k)].y &&
                D[size_t(q)] - cheb(hits[size_t(q)], hits[size_t(k)]) == value[size_t(k)]) predecessors[size_t(k)].push_back(q);
        }
    }
    // Anchors: alone in O at their x, and no optimal link passes over x.  covered[x] > 0: some optimal link passes over ordinal x.
    std::vector<int32_t> covered(static_cast<size_t>(nx) + 2, 0), atX(static_cast<size_t>(nx) + 1, 0);
    auto cover = [&](int32_t from, int32_t to) { if(from + 1 <= to - 1) { covered[size_t(from + 1)] += 1; covered[size_t(to)] -= 1; } };     // ordinals from + 1 .. to - 1
    for(int32_t k = 0; k < n; k++) {
        if(!optimal[size_t(k)]) continue;
        atX[size_t(hits[size_t(k)].x)] += 1;
        if(fromBorder[size_t(k)]) cover(-1, hits[size_t(k)].x);
        if(toBorder[size_t(k)]) cover(hits[size_t(k)].x, int32_t(nx));
        for(const int32_t q : predecessors[size_t(k)]) cover(hits[size_t(q)].x, hits[size_t(k)].x);
    }
    for(size_t x = 1; x < covered.size(); x++) covered[x] += covered[x - 1];
    std::vector<int32_t> anchors;
    for(int32_t k = 0; k < n; k++) {
        if(optimal[size_t(k)] && atX[size_t(hits[size_t(k)].x)] == 1 && covered[size_t(hits[size_t(k)].x)] == 0) anchors.push_back(k);
    }
    r.anchors = anchors.size();
    if(anchors.empty()) { wholeTask(); return; }
    // Number of optimal sub-chains (capped at 2) from `from` (-1: the border) to `to` (n: the border) over the optimal links,
    // and, when there is one, its hits strictly between.
    auto between = [&](int32_t from, int32_t to, std::vector<int32_t>& path) -> uint32_t {
        // ways[k]: optimal sub-chains from `from` to hit k, over hits in (from, to].
        std::vector<uint32_t> ways(count + 1, 0);
        std::vector<int32_t> via(count + 1, -2);
        const int32_t first = from < 0 ? 0 : from + 1, last = to >= n ? n - 1 : to;
        auto reach = [&](int32_t k) {
            uint32_t w = 0; int32_t v = -2;
            if(from < 0 ? fromBorder[size_t(k)] != 0 : false) { w += 1; v = -1; }
            for(const int32_t q : predecessors[size_t(k)]) {
                if(q == from) { w += 1; v = from; }
                else if(q > from && ways[size_t(q)]) { w += ways[size_t(q)]; v = q; }
            }
            ways[size_t(k)] = std::min<uint32_t>(2, w); via[size_t(k)] = v;
        };
        for(int32_t k = first; k <= last; k++) if(optimal[size_t(k)] && (from < 0 || (hits[size_t(k)].x > hits[size_t(from)].x))) reach(k);
        uint32_t total = 0; int32_t end = -2;
        if(to < n) { total = ways[size_t(to)]; end = to; }
        else {
            if(from >= 0 && toBorder[size_t(from)]) { total = 1; end = from; }                 // straight from the anchor to the border
            for(int32_t k = first; k < n; k++) if(optimal[size_t(k)] && toBorder[size_t(k)] && ways[size_t(k)]) { total = std::min<uint32_t>(2, total + ways[size_t(k)]); end = k; }
        }
        path.clear();
        if(total == 1) {
            for(int32_t k = (to < n ? via[size_t(to)] : end); k >= 0 && k != from; k = via[size_t(k)]) path.push_back(k);
            std::reverse(path.begin(), path.end());
        }
        return total;
    };
    // A stretch that holds an ambiguity: the dense DP on the two sub-sequences, padded where an anchor fixes the corner.
    uint64_t synthetic = 1ULL << 40;
    auto window = [&](int32_t from, int32_t to, std::vector< std::pair<uint32_t, uint32_t> >& out) {
        const int32_t x0 = from < 0 ? 0 : hits[size_t(from)].x, y0 = from < 0 ? 0 : hits[size_t(from)].y;            // first marker of each sub-sequence
        const int32_t x1 = to >= n ? int32_t(nx) - 1 : hits[size_t(to)].x, y1 = to >= n ? int32_t(ny) - 1 : hits[size_t(to)].y;      // last
        const int32_t wx = x1 - x0 + 1, wy = y1 - y0 + 1;
        const int32_t pad = std::min(wx, wy) + (wx + wy) / 6 + 2;
        const int32_t padBefore = from < 0 ? 0 : pad, padAfter = to >= n ? 0 : pad;
        std::vector<uint64_t> s0, s1;
        for(int32_t k = 0; k < padBefore; k++) { s0.push_back(synthetic); s1.push_back(synthetic); ++synthetic; }
        for(int32_t x = x0; x <= x1; x++) s0.push_back(uint64_t(seq0[x]));
        for(int32_t y = y0; y <= y1; y++) s1.push_back(uint64_t(seq1[y]));
        for(int32_t k = 0; k < padAfter; k++) { s0.push_back(synthetic); s1.push_back(synthetic); ++synthetic; }
        // Diagonals: x' - y' = (x - x0 + padBefore) - (y - y0 + padBefore) = (x - y) - (x0 - y0).
        const int32_t shift = x0 - y0;
        BandedDpResult dp;
        bandedOverlapAlignment(s0.data(), uint32_t(s0.size()), s1.data(), uint32_t(s1.size()), 6, -1, -1, bandMin - shift, bandMax - shift, dp, policy);
        std::vector< std::pair<uint32_t, uint32_t> > sub;
        diagonalMatches(s0.data(), s1.data(), dp, sub);
        r.denseCells += uint64_t(s0.size()) * uint64_t(int64_t(bandMax) - bandMin + 1);
        r.windows += 1;
        out.clear();
        uint32_t padsSeen = 0;
        for(const auto& p : sub) {
            const int32_t sx = int32_t(p.first) - padBefore, sy = int32_t(p.second) - padBefore;
            if(sx < 0 || sx >= wx) { ++padsSeen; continue; }
            out.push_back(std::make_pair(uint32_t(sx + x0), uint32_t(sy + y0)));
        }
        // Every synthetic marker is aligned with its twin, and the anchors with themselves: else the padding did not hold the corners.
        bool held = padsSeen == uint32_t(padBefore + padAfter);
        if(from >= 0) held = held && !out.empty() && out.front() == std::make_pair(uint32_t(x0), uint32_t(y0));
        if(to < n) held = held && !out.empty() && out.back() == std::make_pair(uint32_t(x1), uint32_t(y1));
        return held;
    };
    std::vector<int32_t> path;
    std::vector< std::pair<uint32_t, uint32_t> > piece;
    bool held = true;
    auto stretch = [&](int32_t from, int32_t to) {
        // Leaves the hits strictly between `from` and `to` in r.ordinals.
        const uint32_t total = between(from, to, path);
        if(total == 1) { for(const int32_t k : path) r.ordinals.push_back(std::make_pair(uint32_t(hits[size_t(k)].x), uint32_t(hits[size_t(k)].y))); return; }
        held = window(from, to, piece) && held;
        for(const auto& p : piece) {
            if(from >= 0 && int32_t(p.first) == hits[size_t(from)].x) continue;
            if(to < n && int32_t(p.first) == hits[size_t(to)].x) continue;
            r.ordinals.push_back(p);
        }
    };
    stretch(-1, anchors.front());
    for(size_t a = 0; a < anchors.size(); a++) {
        r.ordinals.push_back(std::make_pair(uint32_t(hits[size_t(anchors[a])].x), uint32_t(hits[size_t(anchors[a])].y)));
        stretch(anchors[a], a + 1 < anchors.size() ? anchors[a + 1] : n);
    }
    if(!held) { r.ordinals.clear(); r.windows = 0; r.denseCells = 0; wholeTask(); }
}

}  // namespace oracle
#endif
