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
//   findShortestPath                           src/shortestPath.hpp:57-161
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

    // createVertices
    std::vector<std::pair<GraphVertex, size_t>> vertexTable;      // the size_t is CompactUndirectedGraph's edge index, 0 while sorting
    std::array<std::vector<bool>, 2> isLowFrequencyMarker;
    for(size_t i = 0; i < 2; i++) isLowFrequencyMarker[i].assign(markers[i].size(), true);
    {
        auto it0 = markers[0].begin(), it1 = markers[1].begin();
        const auto end0 = markers[0].end(), end1 = markers[1].end();
        while(it0 != end0 && it1 != end1) {
            if(it0->kmerId < it1->kmerId) ++it0;
            else if(it1->kmerId < it0->kmerId) ++it1;
            else {
                const uint32_t kmerId = it0->kmerId;
                auto it0End = it0, it1End = it1;
                while(it0End != end0 && it0End->kmerId == kmerId) ++it0End;
                while(it1End != end1 && it1End->kmerId == kmerId) ++it1End;
                const size_t streakLength0 = size_t(it0End - it0), streakLength1 = size_t(it1End - it1);
                if(streakLength0 > maxMarkerFrequency || streakLength1 > maxMarkerFrequency) {
                    for(auto jt0 = it0; jt0 != it0End; ++jt0) isLowFrequencyMarker[0][jt0->ordinal] = false;
                    for(auto jt1 = it1; jt1 != it1End; ++jt1) isLowFrequencyMarker[1][jt1->ordinal] = false;
                } else {
                    for(auto jt0 = it0; jt0 != it0End; ++jt0) {
                        for(auto jt1 = it1; jt1 != it1End; ++jt1) {
                            GraphVertex vertex;
                            vertex.ordinals = {size_t(jt0->ordinal), size_t(jt1->ordinal)};
                            vertexTable.push_back(std::make_pair(vertex, size_t(0)));
                        }
                    }
                }
                it0 = it0End;
                it1 = it1End;
            }
        }
    }
    std::array<std::vector<uint32_t>, 2> correctedOrdinals;
    for(size_t i = 0; i < 2; i++) {
        correctedOrdinals[i].resize(markers[i].size());
        uint32_t correctedOrdinal = 0;
        for(size_t j = 0; j < markers[i].size(); j++) {
            correctedOrdinals[i][j] = isLowFrequencyMarker[i][j] ? correctedOrdinal++ : std::numeric_limits<uint32_t>::max();
        }
    }

    std::sort(vertexTable.begin(), vertexTable.end());            // sortVertices: pair<vertex, 0> by operator< of pair
    const size_t vStart = vertexTable.size();
    vertexTable.push_back(std::make_pair(GraphVertex(), size_t(0)));
    const size_t vFinish = vertexTable.size();
    vertexTable.push_back(std::make_pair(GraphVertex(), size_t(0)));

    // createEdges
    std::vector<GraphEdge> edgeTable;
    const uint32_t markerCount0 = uint32_t(markers[0].size()), markerCount1 = uint32_t(markers[1].size());
    for(size_t vA = 0; vA < vertexTable.size(); vA++) {
        if(vA == vStart || vA == vFinish) continue;
        const GraphVertex& vertexA = vertexTable[vA].first;
        const int correctedOrdinalA0 = int(correctedOrdinals[0][vertexA.ordinals[0]]);
        const int correctedOrdinalA1 = int(correctedOrdinals[1][vertexA.ordinals[1]]);
        for(size_t vB = vA + 1; vB < vertexTable.size(); vB++) {
            if(vB == vStart || vB == vFinish) continue;
            const GraphVertex& vertexB = vertexTable[vB].first;
            const int correctedOrdinalB0 = int(correctedOrdinals[0][vertexB.ordinals[0]]);
            if(correctedOrdinalB0 > correctedOrdinalA0 + int(maxSkip)) break;
            const int correctedOrdinalB1 = int(correctedOrdinals[1][vertexB.ordinals[1]]);
            if(correctedOrdinalB1 < correctedOrdinalA1) continue;
            if(size_t(std::abs(correctedOrdinalB1 - correctedOrdinalA1)) > maxSkip) continue;
            if(maxDrift < maxSkip) {
                const int offsetA = correctedOrdinalA0 - correctedOrdinalA1;
                const int offsetB = correctedOrdinalB0 - correctedOrdinalB1;
                if(size_t(std::abs(offsetA - offsetB)) > maxDrift) continue;
            }
            const int delta0 = correctedOrdinalB0 - correctedOrdinalA0;
            const int delta1 = correctedOrdinalB1 - correctedOrdinalA1;
            const size_t weight = size_t(std::abs(delta0 - 1) + std::abs(delta1 - 1));
            edgeTable.push_back(GraphEdge{{vA, vB}, weight});
        }
    }
    for(size_t v = 0; v < vertexTable.size(); v++) {
        if(v == vStart || v == vFinish) continue;
        const GraphVertex& vertex = vertexTable[v].first;
        const int correctedOrdinal0 = int(correctedOrdinals[0][vertex.ordinals[0]]);
        const int correctedOrdinal1 = int(correctedOrdinals[1][vertex.ordinals[1]]);
        const int deltaFinish0 = int(markerCount0) - correctedOrdinal0;
        const int deltaFinish1 = int(markerCount1) - correctedOrdinal1;
        edgeTable.push_back(GraphEdge{{v, vStart}, uint64_t(std::abs(correctedOrdinal0) + std::abs(correctedOrdinal1))});
        edgeTable.push_back(GraphEdge{{v, vFinish}, uint64_t(std::abs(deltaFinish0) + std::abs(deltaFinish1))});
    }

    // doneAddingEdges: degree count, running sum, fill backwards, reverse every list.
    for(const GraphEdge& e : edgeTable) { ++vertexTable[e.vertices[0]].second; ++vertexTable[e.vertices[1]].second; }
    size_t total = 0;
    for(auto& p : vertexTable) { total += p.second; p.second = total; }
    vertexTable.push_back(std::make_pair(GraphVertex(), total));
    std::vector<size_t> edgeLists(total);
    for(size_t e = 0; e < edgeTable.size(); e++) {
        edgeLists[--vertexTable[edgeTable[e].vertices[0]].second] = e;
        edgeLists[--vertexTable[edgeTable[e].vertices[1]].second] = e;
    }
    for(size_t v = 0; v + 1 < vertexTable.size(); v++) {
        std::reverse(edgeLists.begin() + vertexTable[v].second, edgeLists.begin() + vertexTable[v + 1].second);
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
    std::vector<size_t> shortestPath;
    while(!q.empty()) {
        const auto p0 = q.top();
        q.pop();
        const uint64_t distance0 = p0.first;
        const size_t v0 = p0.second;
        GraphVertex& vertex0 = vertexTable[v0].first;
        if(vertex0.color == 1) continue;
        vertex0.color = 1;
        if(v0 == vFinish) {
            size_t v = v0;
            while(true) {
                shortestPath.push_back(v);
                if(v == vStart) break;
                v = vertexTable[v].first.predecessor;
            }
            std::reverse(shortestPath.begin(), shortestPath.end());
            break;
        }
        for(size_t k = vertexTable[v0].second; k != vertexTable[v0 + 1].second; k++) {
            const GraphEdge& e01 = edgeTable[edgeLists[k]];
            const size_t v1 = e01.vertices[0] == v0 ? e01.vertices[1] : e01.vertices[0];
            GraphVertex& vertex1 = vertexTable[v1].first;
            if(vertex1.color == 1) continue;
            const uint64_t distance1 = distance0 + e01.weight;
            if(distance1 < vertex1.distance) {
                q.push(std::make_pair(distance1, v1));
                vertex1.predecessor = v0;
                vertex1.distance = distance1;
            }
        }
    }
    for(const size_t v : shortestPath) {
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
