// TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's alignment method 0 as
// Assembler::flagPalindromicReads uses it.  Checked against the reference's
// own AlignmentGraph.cpp in oracle/_ref (tests/test_oracle_vs_ref.py) and against fixtures it made
// (tests/golden/palindromic.npz).  Nothing in the product includes this file.
//
// Follows, structure for structure:
//   Assembler::getMarkersSortedByKmerId        src/AssemblerMarkers.cpp:83-98
//   AlignmentGraph::create                     src/AlignmentGraph.cpp:58-133
//   AlignmentGraph::createVertices             src/AlignmentGraph.cpp:155-263
//   CompactUndirectedGraph::sortVertices       src/CompactUndirectedGraph.hpp:503-510
//   AlignmentGraph::createEdges                src/AlignmentGraph.cpp:289-395
//   CompactUndirectedGraph::doneAddingEdges    src/CompactUndirectedGraph.hpp:536-582
//   findShortestPath                           src/route.hpp:57-161
//   Assembler::flagPalindromicReadsThreadFunction   src/AssemblerAlign.cpp:702-770
// The result of the reference depends on how std::sort orders equal keys (twice) and on how
// std::priority_queue orders equal distances; the restatement therefore makes the same standard-library
// calls on sequences that compare the same way (parity holds against a reference built with the same
// C++ library, which oracle/_ref is).
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <queue>
#include <utility>
#include <vector>

namespace oracle {
namespace method0 {

struct MarkerWithOrdinal {                     // src/Marker.hpp:99-117
    uint32_t kmerId;
    uint32_t ordinal;
    bool operator<(const MarkerWithOrdinal& that) const { return kmerId < that.kmerId; }
};

struct GraphVertex {                           // src/AlignmentGraph.hpp:84-109 (the members the search reads)
    std::array<size_t, 2> ordinals;
    size_t predecessor;
    uint64_t distance;
    uint8_t color;
    bool operator<(const GraphVertex& that) const { return ordinals[0] < that.ordinals[0]; }
};

struct GraphEdge { std::array<size_t, 2> vertices; uint64_t weight; };

struct OrderByDistanceGreater {                // src/orderPairs.hpp:35-42
    bool operator()(const std::pair<uint64_t, size_t>& x, const std::pair<uint64_t, size_t>& y) const { return x.first > y.first; }
};

// The alignment of read strand 0 (kmerIds[0]) with strand 1 (kmerIds[1]): (ordinal0, ordinal1) per vertex of the path.
inline void align(const std::array<std::vector<uint32_t>, 2>& kmerIds, size_t maxSkip, size_t maxDrift, uint32_t maxMarkerFrequency,
    std::vector<std::array<uint32_t, 2>>& alignment)
{
    alignment.clear();
    std::array<std::vector<MarkerWithOrdinal>, 2> markers;
    for(size_t s = 0; s < 2; s++) {
        markers[s].resize(kmerIds[s].size());
        for(uint32_t ordinal = 0; ordinal < kmerIds[s].size(); ordinal++) markers[s][ordinal] = MarkerWithOrdinal{kmerIds[s][ordinal], ordinal};
        std::sort(markers[s].begin(), markers[s].end());
    }

    // createVertices: one vertex per pair of markers with the same kmer id, unless that kmer id occurs
    // more than maxMarkerFrequency times on either strand (then its markers stop counting as markers).
    std::vector<std::pair<GraphVertex, size_t>> vertexTable;      // the size_t is CompactUndirectedGraph's edge index, 0 while sorting
    std::array<std::vector<bool>, 2> keepsItsOrdinal;
    for(size_t s = 0; s < 2; s++) keepsItsOrdinal[s].assign(markers[s].size(), true);
    {
        const std::vector<MarkerWithOrdinal>& m0 = markers[0];
        const std::vector<MarkerWithOrdinal>& m1 = markers[1];
        size_t p0 = 0, p1 = 0;
        while(p0 < m0.size() && p1 < m1.size()) {
            if(m0[p0].kmerId < m1[p1].kmerId) { ++p0; continue; }
            if(m1[p1].kmerId < m0[p0].kmerId) { ++p1; continue; }
            const uint32_t common = m0[p0].kmerId;
            size_t q0 = p0, q1 = p1;                              // ends of the two runs of this kmer id
            while(q0 < m0.size() && m0[q0].kmerId == common) ++q0;
            while(q1 < m1.size() && m1[q1].kmerId == common) ++q1;
            if(q0 - p0 > maxMarkerFrequency || q1 - p1 > maxMarkerFrequency) {
                for(size_t k = p0; k < q0; k++) keepsItsOrdinal[0][m0[k].ordinal] = false;
                for(size_t k = p1; k < q1; k++) keepsItsOrdinal[1][m1[k].ordinal] = false;
            } else {
                for(size_t k0 = p0; k0 < q0; k0++) {
                    for(size_t k1 = p1; k1 < q1; k1++) {
                        GraphVertex vertex;
                        vertex.ordinals = {size_t(m0[k0].ordinal), size_t(m1[k1].ordinal)};
                        vertexTable.push_back(std::make_pair(vertex, size_t(0)));
                    }
                }
            }
            p0 = q0;
            p1 = q1;
        }
    }
    std::array<std::vector<uint32_t>, 2> rank;
    for(size_t i = 0; i < 2; i++) {
        rank[i].resize(markers[i].size());
        uint32_t nextRank = 0;
        for(size_t j = 0; j < markers[i].size(); j++) {
            rank[i][j] = keepsItsOrdinal[i][j] ? nextRank++ : std::numeric_limits<uint32_t>::max();
        }
    }

    std::sort(vertexTable.begin(), vertexTable.end());            // sortVertices: pair<vertex, 0> by operator< of pair
    const size_t vStart = vertexTable.size();
    vertexTable.push_back(std::make_pair(GraphVertex(), size_t(0)));
    const size_t vFinish = vertexTable.size();
    vertexTable.push_back(std::make_pair(GraphVertex(), size_t(0)));

    // createEdges
    std::vector<GraphEdge> edges;
    const uint32_t markerCount0 = uint32_t(markers[0].size()), markerCount1 = uint32_t(markers[1].size());
    for(size_t vA = 0; vA < vertexTable.size(); vA++) {
        if(vA == vStart || vA == vFinish) continue;
        const GraphVertex& from = vertexTable[vA].first;
        const int a0 = int(rank[0][from.ordinals[0]]);
        const int a1 = int(rank[1][from.ordinals[1]]);
        for(size_t vB = vA + 1; vB < vertexTable.size(); vB++) {
            if(vB == vStart || vB == vFinish) continue;
            const GraphVertex& to = vertexTable[vB].first;
            const int b0 = int(rank[0][to.ordinals[0]]);
            if(b0 > a0 + int(maxSkip)) break;
            const int b1 = int(rank[1][to.ordinals[1]]);
            if(b1 < a1) continue;
            if(size_t(std::abs(b1 - a1)) > maxSkip) continue;
            if(maxDrift < maxSkip) {
                const int diagonalA = a0 - a1;
                const int diagonalB = b0 - b1;
                if(size_t(std::abs(diagonalA - diagonalB)) > maxDrift) continue;
            }
            const int step0 = b0 - a0;
            const int step1 = b1 - a1;
            const size_t weight = size_t(std::abs(step0 - 1) + std::abs(step1 - 1));
            edges.push_back(GraphEdge{{vA, vB}, weight});
        }
    }
    for(size_t v = 0; v < vertexTable.size(); v++) {
        if(v == vStart || v == vFinish) continue;
        const GraphVertex& vertex = vertexTable[v].first;
        const int c0 = int(rank[0][vertex.ordinals[0]]);
        const int c1 = int(rank[1][vertex.ordinals[1]]);
        const int toEnd0 = int(markerCount0) - c0;
        const int toEnd1 = int(markerCount1) - c1;
        edges.push_back(GraphEdge{{v, vStart}, uint64_t(std::abs(c0) + std::abs(c1))});
        edges.push_back(GraphEdge{{v, vFinish}, uint64_t(std::abs(toEnd0) + std::abs(toEnd1))});
    }

    // doneAddingEdges: degree count, running sum, fill backwards, reverse every list.
    for(const GraphEdge& e : edges) { ++vertexTable[e.vertices[0]].second; ++vertexTable[e.vertices[1]].second; }
    size_t total = 0;
    for(auto& p : vertexTable) { total += p.second; p.second = total; }
    vertexTable.push_back(std::make_pair(GraphVertex(), total));
    std::vector<size_t> incidence(total);
    for(size_t e = 0; e < edges.size(); e++) {
        incidence[--vertexTable[edges[e].vertices[0]].second] = e;
        incidence[--vertexTable[edges[e].vertices[1]].second] = e;
    }
    for(size_t v = 0; v + 1 < vertexTable.size(); v++) {
        std::reverse(incidence.begin() + vertexTable[v].second, incidence.begin() + vertexTable[v + 1].second);
    }
    const size_t vertexCount = vertexTable.size() - 1;

    // findShortestPath
    const size_t nullVertex = std::numeric_limits<size_t>::max();
    for(size_t v = 0; v < vertexCount; v++) {
        GraphVertex& vertex = vertexTable[v].first;
        vertex.predecessor = nullVertex;
        vertex.distance = std::numeric_limits<uint64_t>::max();
        vertex.color = 0;
    }
    vertexTable[vStart].first.predecessor = vStart;
    vertexTable[vStart].first.distance = 0;
    std::priority_queue<std::pair<uint64_t, size_t>, std::vector<std::pair<uint64_t, size_t>>, OrderByDistanceGreater> q;
    q.push(std::make_pair(uint64_t(0), vStart));
    std::vector<size_t> route;
    while(!q.empty()) {
        const auto p0 = q.top();
        q.pop();
        const uint64_t reached = p0.first;
        const size_t v0 = p0.second;
        GraphVertex& here = vertexTable[v0].first;
        if(here.color == 1) continue;
        here.color = 1;
        if(v0 == vFinish) {
            size_t v = v0;
            while(true) {
                route.push_back(v);
                if(v == vStart) break;
                v = vertexTable[v].first.predecessor;
            }
            std::reverse(route.begin(), route.end());
            break;
        }
        for(size_t k = vertexTable[v0].second; k != vertexTable[v0 + 1].second; k++) {
            const GraphEdge& link = edges[incidence[k]];
            const size_t v1 = link.vertices[0] == v0 ? link.vertices[1] : link.vertices[0];
            GraphVertex& there = vertexTable[v1].first;
            if(there.color == 1) continue;
            const uint64_t candidate = reached + link.weight;
            if(candidate < there.distance) {
                q.push(std::make_pair(candidate, v1));
                there.predecessor = v0;
                there.distance = candidate;
            }
        }
    }
    for(const size_t v : route) {
        if(v == vStart || v == vFinish) continue;
        const GraphVertex& vertex = vertexTable[v].first;
        alignment.push_back({uint32_t(vertex.ordinals[0]), uint32_t(vertex.ordinals[1])});
    }
}

struct ReadVerdict { bool palindromic; uint32_t alignedMarkerCount, nearDiagonalMarkerCount; };

// The body of the loop over reads, src/AssemblerAlign.cpp:720-756.
inline ReadVerdict flagRead(const std::array<std::vector<uint32_t>, 2>& kmerIds, uint32_t maxSkip, uint32_t maxDrift, uint32_t maxMarkerFrequency,
    double alignedFractionThreshold, double nearDiagonalFractionThreshold, uint32_t deltaThreshold,
    std::vector<std::array<uint32_t, 2>>& alignment)
{
    align(kmerIds, maxSkip, maxDrift, maxMarkerFrequency, alignment);
    ReadVerdict verdict{false, uint32_t(alignment.size()), 0};
    const size_t totalMarkerCount = kmerIds[0].size();
    for(const auto& ordinals : alignment) {
        const uint32_t delta = uint32_t(std::abs(int32_t(ordinals[0]) - int32_t(ordinals[1])));
        if(delta < deltaThreshold) verdict.nearDiagonalMarkerCount++;
    }
    const double alignedFraction = double(alignment.size()) / double(totalMarkerCount);
    if(alignedFraction < alignedFractionThreshold) return verdict;
    const double nearDiagonalFraction = double(verdict.nearDiagonalMarkerCount) / double(totalMarkerCount);
    if(nearDiagonalFraction < nearDiagonalFractionThreshold) return verdict;
    verdict.palindromic = true;
    return verdict;
}

// FNV-1a over the ordinals of an alignment: lets two implementations compare whole alignments through one number.
inline uint64_t digest(const std::vector<std::array<uint32_t, 2>>& alignment)
{
    uint64_t h = 1469598103934665603ULL;
    for(const auto& p : alignment) for(const uint32_t v : p) for(int b = 0; b < 4; b++) { h ^= (v >> (8 * b)) & 0xffu; h *= 1099511628211ULL; }
    return h;
}

}  // namespace method0
}  // namespace oracle
