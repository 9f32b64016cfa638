// TEST INFRASTRUCTURE ONLY -- nothing under shasta_amd/ may include, link or call this.
//
// The banded overlap alignment of /root/reference/src/Align4.cpp:993-1088 (6 / -1 / -1, all end gaps free), computed from the
// MATCHES inside the band instead of from every cell -- the CPU statement of the device's sparse path (K10s), and the thing
// that was costed against the dense DP (oracle/banded_dp.hpp) before any kernel was written.
//
// Why it is exact.  Only aligned pairs of EQUAL markers leave computeBandedAlignment (:1053-1068).  Write a path's score as
// 6 M - X - G (M matches, X mismatched diagonal steps, G gap steps).  Between two consecutive matches m' = (x', y') and
// m = (x, y), x' < x, y' < y, the best that a stretch without a match can do is a Chebyshev walk: min(dx, dy) mismatched
// diagonals and |dx - dy| gaps, dx = x - x' - 1, dy = y - y' - 1, cost max(dx, dy); it stays inside the band (its diagonals lie
// between those of its ends) and inside the matrix.  From the free border to a match the cheapest way in is the match's own
// diagonal, cost min(x, y); from a match to the free border likewise, min(nx - 1 - x, ny - 1 - y).  Hence, over the matches in
// the band ("hits"), sorted by x:
//
//     D(m) = 6 + max( -min(x, y),  max over hits m' with x' < x, y' < y of  D(m') - max(x - x' - 1, y - y' - 1) )
//     best = max over m of  D(m) - min(nx - 1 - x, ny - 1 - y),      against the best path WITHOUT a match, Z (0 or negative)
//
// equals the dense optimum (a walk that crosses a further match diagonally scores MORE than the formula says, and the chain that
// contains that match accounts for it).  The dense traceback's output is the match set of SOME optimal path, whichever the tie
// policy; if the optimal CHAIN (the set of matches) is unique, every policy -- every reading of SeqAn -- gives exactly that set.
// So the sparse path counts optimal chains (capped at 2) and answers only when there is exactly one ("certified"); otherwise, or
// when a hit's scan for predecessors runs long (repeats), the task goes to the dense DP.  A certified answer does not depend on
// the tie policy at all.
//
// The scan for predecessors walks back over the hits before m in x order and stops at the first k where
// prefixMax(D)[k] - (x - x_k - 1) < best so far (no earlier hit can reach `best`; `<`, not `<=`: ties must be seen to be counted).
#ifndef ORACLE_SPARSE_CHAIN_HPP
#define ORACLE_SPARSE_CHAIN_HPP

#include <algorithm>
#include <cstdint>
#include <limits>
#include <unordered_map>
#include <utility>
#include <vector>

namespace oracle {

struct SparseChainResult {
    bool certified = false;            // exactly one optimal chain (or none at all: the empty alignment), found within the scan budget
    int32_t score = std::numeric_limits<int32_t>::min();
    std::vector< std::pair<uint32_t, uint32_t> > ordinals;     // the chain, ascending
    uint64_t hits = 0, scanSteps = 0;
    int reason = 0;                    // 0 certified, 1 several optimal chains, 2 a chain ties with the empty alignment, 3 scan budget, 4 two hits of one marker
};

// The best score of a path without any match: the largest -min(i, j) over the border cells (i = nx or j = ny) inside the band.
inline int32_t bestMatchlessScore(uint32_t nx, uint32_t ny, int32_t bandMin, int32_t bandMax)
{
    int64_t best = std::numeric_limits<int32_t>::min();
    // Column i = nx: rows j with bandMin <= nx - j <= bandMax; -min(nx, j) is largest at the smallest j.
    {
        const int64_t jLo = std::max<int64_t>(0, int64_t(nx) - bandMax), jHi = std::min<int64_t>(ny, int64_t(nx) - bandMin);
        if(jLo <= jHi) best = std::max(best, -std::min<int64_t>(nx, jLo));
    }
    // Row j = ny: columns i with bandMin <= i - ny <= bandMax; smallest i.
    {
        const int64_t iLo = std::max<int64_t>(0, int64_t(ny) + bandMin), iHi = std::min<int64_t>(nx, int64_t(ny) + bandMax);
        if(iLo <= iHi) best = std::max(best, -std::min<int64_t>(ny, iLo));
    }
    return int32_t(best);
}

template<class T>
inline void sparseChainAlignment(const T* seq0, uint32_t nx, const T* seq1, uint32_t ny, int32_t bandMin, int32_t bandMax,
    SparseChainResult& r, uint32_t scanBudgetPerHit = 64, bool oneHitPerMarker = false, bool runningMaxBound = false)
{
    r = SparseChainResult();
    if(bandMin > bandMax || bandMin > int32_t(nx) || bandMax < -int32_t(ny)) return;       // (the dense DP fails there as well: not a task)
    // Hits inside the band, by x then y.
    struct Hit { uint32_t x, y; };
    std::vector<Hit> hits;
    {
        std::unordered_map<T, std::vector<uint32_t> > where;
        for(uint32_t y = 0; y < ny; y++) where[seq1[y]].push_back(y);
        for(uint32_t x = 0; x < nx; x++) {
            const auto it = where.find(seq0[x]);
            if(it == where.end()) continue;
            for(const uint32_t y : it->second) {
                const int64_t d = int64_t(x) - int64_t(y);
                if(d >= bandMin && d <= bandMax) hits.push_back(Hit{x, y});
            }
        }
    }
    const size_t n = hits.size();
    r.hits = n;
    // (The device ranks a task's hits by their ordinal in one of the two reads with a bit per marker: two hits of one marker of
    // that read inside the band send the task to the dense DP.  oneHitPerMarker restates that, for read 0.)
    if(oneHitPerMarker) for(size_t k = 1; k < n; k++) if(hits[k].x == hits[k - 1].x) { r.reason = 4; return; }
    std::vector<int32_t> D(n), prefixMax(n);
    std::vector<int32_t> pred(n);
    std::vector<uint8_t> ways(n);          // optimal chains ending here, capped at 2
    const int32_t Z = bestMatchlessScore(nx, ny, bandMin, bandMax);
    int32_t best = std::numeric_limits<int32_t>::min();
    int64_t bestAt = -1;
    uint32_t bestWays = 0;
    bool overBudget = false;
    for(size_t k = 0; k < n; k++) {
        const int32_t x = int32_t(hits[k].x), y = int32_t(hits[k].y);
        int32_t value = -std::min(x, y);
        int32_t from = -1;
        uint32_t count = 1;
        uint32_t steps = 0;
        for(int64_t q = int64_t(k) - 1; q >= 0; q--) {
            const int32_t xq = int32_t(hits[size_t(q)].x), yq = int32_t(hits[size_t(q)].y);
            // (runningMaxBound: the device's form of the bound -- the largest D of ALL hits before m, which it has in a register,
            // in place of the prefix maximum up to q, which it would have to keep per hit: weaker, never wrong.)
            if((runningMaxBound ? prefixMax[k - 1] : prefixMax[size_t(q)]) - (x - xq - 1) < value) break;
            if(++steps > scanBudgetPerHit) { overBudget = true; break; }
            if(xq >= x || yq >= y) continue;
            const int32_t candidate = D[size_t(q)] - std::max(x - xq - 1, y - yq - 1);
            if(candidate > value) { value = candidate; from = int32_t(q); count = ways[size_t(q)]; }
            else if(candidate == value) count = std::min<uint32_t>(2, count + ways[size_t(q)]);
        }
        r.scanSteps += steps;
        D[k] = 6 + value; pred[k] = from; ways[k] = uint8_t(std::min<uint32_t>(2, count));
        prefixMax[k] = k ? std::max(prefixMax[k - 1], D[k]) : D[k];
        const int32_t end = D[k] - std::min(int32_t(nx) - 1 - x, int32_t(ny) - 1 - y);
        if(end > best) { best = end; bestAt = int64_t(k); bestWays = ways[k]; }
        else if(end == best) bestWays = std::min<uint32_t>(2, bestWays + ways[k]);
    }
    if(overBudget) { r.reason = 3; return; }
    if(n == 0 || best < Z) { r.certified = true; r.score = Z; return; }            // every optimal path is without a match: nothing is aligned
    if(best == Z) { r.reason = 2; return; }
    if(bestWays != 1) { r.reason = 1; return; }
    r.certified = true; r.score = best;
    for(int64_t k = bestAt; k >= 0; k = pred[size_t(k)]) r.ordinals.push_back(std::make_pair(hits[size_t(k)].x, hits[size_t(k)].y));
    std::reverse(r.ordinals.begin(), r.ordinals.end());
}

}  // namespace oracle
#endif
