#include "PalindromicReads.hpp"
#include "OverlapStages.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <thread>

namespace shasta_mi355x {
namespace host {

namespace {

// A marker as method 0 sees it: ordered by kmer id ONLY (src/Marker.hpp:112-116), so the order in
// which std::sort leaves equal kmer ids is part of the result.
struct SortedMarker {
    uint32_t kmerId, ordinal;
    bool operator<(const SortedMarker& that) const { return kmerId < that.kmerId; }
};

// A vertex = a pair of markers with the same kmer id; ordered by the first ordinal ONLY
// (src/AlignmentGraph.hpp:105-108; the graph sorts pair<vertex, 0>, whose second members are all equal).
struct Vertex {
    uint32_t ordinal0, ordinal1;
    bool operator<(const Vertex& that) const { return ordinal0 < that.ordinal0; }
};

struct Edge { uint32_t v0, v1; uint64_t weight; };

// Work areas of one thread, reused from read to read.
struct Method0 {
    std::vector<SortedMarker> sorted[2];
    std::vector<char> lowFrequency[2];
    std::vector<uint32_t> corrected[2];
    std::vector<Vertex> vertices;
    std::vector<Edge> edges;
    std::vector<uint64_t> firstEdge, edgeList;             // CSR of edge ids per vertex, ascending
    std::vector<uint64_t> distance;
    std::vector<uint32_t> predecessor;
    std::vector<char> done;
    using QueueEntry = std::pair<uint64_t, uint32_t>;      // (distance, vertex); ordered by distance ONLY (src/orderPairs.hpp:35-42)
    struct FartherFirst { bool operator()(const QueueEntry& x, const QueueEntry& y) const { return x.first > y.first; } };
    std::vector<uint32_t> path;

    void align(const uint32_t* k0, const uint32_t* k1, uint32_t n, uint32_t maxSkip, uint32_t maxDrift, uint32_t maxMarkerFrequency,
        std::vector<std::pair<uint32_t, uint32_t>>& alignment);
};

void Method0::align(const uint32_t* k0, const uint32_t* k1, uint32_t n, uint32_t maxSkip, uint32_t maxDrift, uint32_t maxMarkerFrequency,
    std::vector<std::pair<uint32_t, uint32_t>>& alignment)
{
    alignment.clear();
    const uint32_t* kmerIds[2] = {k0, k1};
    // Assembler::getMarkersSortedByKmerId, src/AssemblerMarkers.cpp:83-98.
    for(int s = 0; s < 2; s++) {
        sorted[s].resize(n);
        for(uint32_t ordinal = 0; ordinal < n; ordinal++) sorted[s][ordinal] = SortedMarker{kmerIds[s][ordinal], ordinal};
        std::sort(sorted[s].begin(), sorted[s].end());
        lowFrequency[s].assign(n, 1);
    }

    // AlignmentGraph::createVertices, src/AlignmentGraph.cpp:155-263: joint sweep over the two sorted lists; a
    // kmer id whose streak is longer than maxMarkerFrequency on either side gives no vertices and its markers
    // stop counting as markers; every other common kmer id gives streak0 x streak1 vertices, in streak order.
    vertices.clear();
    {
        size_t i0 = 0, i1 = 0;
        while(i0 != n && i1 != n) {
            if(sorted[0][i0].kmerId < sorted[1][i1].kmerId) { ++i0; continue; }
            if(sorted[1][i1].kmerId < sorted[0][i0].kmerId) { ++i1; continue; }
            const uint32_t kmerId = sorted[0][i0].kmerId;
            size_t e0 = i0, e1 = i1;
            while(e0 != n && sorted[0][e0].kmerId == kmerId) ++e0;
            while(e1 != n && sorted[1][e1].kmerId == kmerId) ++e1;
            if(e0 - i0 > maxMarkerFrequency || e1 - i1 > maxMarkerFrequency) {
                for(size_t j = i0; j != e0; ++j) lowFrequency[0][sorted[0][j].ordinal] = 0;
                for(size_t j = i1; j != e1; ++j) lowFrequency[1][sorted[1][j].ordinal] = 0;
            } else {
                for(size_t j0 = i0; j0 != e0; ++j0) {
                    for(size_t j1 = i1; j1 != e1; ++j1) vertices.push_back(Vertex{sorted[0][j0].ordinal, sorted[1][j1].ordinal});
                }
            }
            i0 = e0; i1 = e1;
        }
    }
    // Ordinals that count low-frequency markers only (:253-263).
    for(int s = 0; s < 2; s++) {
        corrected[s].resize(n);
        uint32_t next = 0;
        for(uint32_t j = 0; j < n; j++) corrected[s][j] = lowFrequency[s][j] ? next++ : std::numeric_limits<uint32_t>::max();
    }
    std::sort(vertices.begin(), vertices.end());                                     // sortVertices, :78
    const uint32_t markerVertexCount = uint32_t(vertices.size());
    const uint32_t vStart = markerVertexCount, vFinish = markerVertexCount + 1;     // :83-84
    const uint32_t vertexCount = markerVertexCount + 2;

    // createEdges, :289-395.  Signed 32-bit arithmetic and the unsigned comparisons exactly as written there:
    // `abs(int) > size_t` converts the int to size_t.
    edges.clear();
    for(uint32_t a = 0; a < markerVertexCount; a++) {
        const int a0 = int(corrected[0][vertices[a].ordinal0]), a1 = int(corrected[1][vertices[a].ordinal1]);
        for(uint32_t b = a + 1; b < markerVertexCount; b++) {
            const int b0 = int(corrected[0][vertices[b].ordinal0]);
            if(b0 > a0 + int(maxSkip)) break;                                       // :336
            const int b1 = int(corrected[1][vertices[b].ordinal1]);
            if(b1 < a1) continue;                                                   // :344
            if(size_t(std::abs(b1 - a1)) > size_t(maxSkip)) continue;                // :349
            if(maxDrift < maxSkip) {                                                // :354-360
                const int offsetA = a0 - a1, offsetB = b0 - b1;
                if(size_t(std::abs(offsetA - offsetB)) > size_t(maxDrift)) continue;
            }
            const int delta0 = b0 - a0, delta1 = b1 - a1;
            edges.push_back(Edge{a, b, uint64_t(size_t(std::abs(delta0 - 1) + std::abs(delta1 - 1)))});   // :368-369
        }
    }
    for(uint32_t v = 0; v < markerVertexCount; v++) {                               // :374-393
        const int c0 = int(corrected[0][vertices[v].ordinal0]), c1 = int(corrected[1][vertices[v].ordinal1]);
        const int f0 = int(n) - c0, f1 = int(n) - c1;
        edges.push_back(Edge{v, vStart, uint64_t(std::abs(c0) + std::abs(c1))});
        edges.push_back(Edge{v, vFinish, uint64_t(std::abs(f0) + std::abs(f1))});
    }
    // CompactUndirectedGraph::doneAddingEdges, src/CompactUndirectedGraph.hpp:536-582: per vertex, the ids of
    // its edges in ascending order.
    firstEdge.assign(size_t(vertexCount) + 1, 0);
    for(const Edge& e : edges) { ++firstEdge[e.v0 + 1]; ++firstEdge[e.v1 + 1]; }
    for(uint32_t v = 0; v < vertexCount; v++) firstEdge[v + 1] += firstEdge[v];
    edgeList.resize(2 * edges.size());
    {
        std::vector<uint64_t>& fill = distance;                                     // scratch
        fill.assign(firstEdge.begin(), firstEdge.end() - 1);
        for(uint64_t e = 0; e < edges.size(); e++) { edgeList[fill[edges[e].v0]++] = e; edgeList[fill[edges[e].v1]++] = e; }
    }

    // findShortestPath, src/shortestPath.hpp:57-161 (lazy deletion; not a textbook Dijkstra, as the file says).
    distance.assign(vertexCount, std::numeric_limits<uint64_t>::max());
    predecessor.assign(vertexCount, std::numeric_limits<uint32_t>::max());
    done.assign(vertexCount, 0);
    predecessor[vStart] = vStart;
    distance[vStart] = 0;
    std::priority_queue<QueueEntry, std::vector<QueueEntry>, FartherFirst> q;
    q.push(QueueEntry(0, vStart));
    path.clear();
    while(!q.empty()) {
        const QueueEntry top = q.top();
        q.pop();
        const uint64_t distance0 = top.first;
        const uint32_t v0 = top.second;
        if(done[v0]) continue;
        done[v0] = 1;
        if(v0 == vFinish) {
            for(uint32_t v = v0;; v = predecessor[v]) { path.push_back(v); if(v == vStart) break; }
            std::reverse(path.begin(), path.end());
            break;
        }
        for(uint64_t k = firstEdge[v0]; k != firstEdge[v0 + 1]; k++) {
            const Edge& e = edges[edgeList[k]];
            const uint32_t v1 = e.v0 == v0 ? e.v1 : e.v0;
            if(done[v1]) continue;
            const uint64_t distance1 = distance0 + e.weight;
            if(distance1 < distance[v1]) {
                q.push(QueueEntry(distance1, v1));
                predecessor[v1] = v0;
                distance[v1] = distance1;
            }
        }
    }
    for(const uint32_t v : path) {                                                   // :118-126
        if(v == vStart || v == vFinish) continue;
        alignment.push_back(std::make_pair(vertices[v].ordinal0, vertices[v].ordinal1));
    }
}

inline uint32_t kmerIdOf(const CompressedMarker7& m)
{
    return uint32_t(m.bytes[0]) | (uint32_t(m.bytes[1]) << 8) | (uint32_t(m.bytes[2]) << 16) | (uint32_t(m.bytes[3]) << 24);
}

}  // namespace

void selfAlignmentMethod0(const uint32_t* kmerIds0, const uint32_t* kmerIds1, uint32_t n,
    uint32_t maxSkip, uint32_t maxDrift, uint32_t maxMarkerFrequency, std::vector<std::pair<uint32_t, uint32_t>>& alignment)
{
    Method0 work;
    work.align(kmerIds0, kmerIds1, n, maxSkip, maxDrift, maxMarkerFrequency, alignment);
}

PalindromicReadCounts flagPalindromicReads(const std::string& dataDirectory, const PalindromicReadOptions& o, size_t threadCount)
{
    if(threadCount == 0) threadCount = std::thread::hardware_concurrency();         // :664-666
    if(threadCount == 0) threadCount = 1;
    Markers markers;
    markers.accessExistingReadOnly(dataDirectory + "/Markers");
    ReadFlagsVector flags;
    flags.accessExistingReadWrite(dataDirectory + "/ReadFlags");
    const uint64_t readCount = markers.size() / 2;
    if(flags.size() != readCount) throw std::runtime_error("flagPalindromicReads: Data/ReadFlags and Data/Markers disagree on the number of reads.");   // :677
    for(uint64_t r = 0; r < readCount; r++) flags[r] = uint8_t(flags[r] & ~1u);      // :678-681

    // Device: the bound for every read.
    std::vector<uint32_t> bound(readCount, 0);
    if(readCount) {
        shasta_mi355x_ctx* ctx = shasta_mi355x_create(0);
        if(!ctx) throw std::runtime_error(shasta_mi355x_last_error());
        const bool failed = shasta_mi355x_set_markers(ctx, readCount, markers.toc.begin(), markers.data.begin(), nullptr) ||
            shasta_mi355x_palindromic_screen(ctx, o.deltaThreshold, bound.data());
        const std::string message = failed ? shasta_mi355x_last_error() : "";
        shasta_mi355x_destroy(ctx);
        if(failed) throw std::runtime_error(message);
    }

    // Host: the reads the bound does not settle, by the reference's procedure.
    PalindromicReadCounts counts;
    counts.readCount = readCount;
    std::vector<uint64_t> survivors;
    for(uint64_t r = 0; r < readCount; r++) {
        const double n = double(markers.toc[2 * r + 1] - markers.toc[2 * r]);
        if(double(bound[r]) / n < o.nearDiagonalFractionThreshold) ++counts.screenedOut;     // cannot pass :750
        else survivors.push_back(r);
    }
    std::atomic<uint64_t> next(0);
    std::string firstError;
    std::mutex errorMutex;
    auto worker = [&]() {
        try {
            Method0 work;
            std::vector<uint32_t> k0, k1;
            std::vector<std::pair<uint32_t, uint32_t>> alignment;
            for(;;) {
                const uint64_t s = next.fetch_add(1);
                if(s >= survivors.size()) break;
                const uint64_t r = survivors[s];
                const uint32_t n = uint32_t(markers.toc[2 * r + 1] - markers.toc[2 * r]);
                k0.resize(n); k1.resize(n);
                for(uint32_t i = 0; i < n; i++) { k0[i] = kmerIdOf(markers.begin(2 * r)[i]); k1[i] = kmerIdOf(markers.begin(2 * r + 1)[i]); }
                work.align(k0.data(), k1.data(), n, o.maxSkip, o.maxDrift, o.maxMarkerFrequency, alignment);
                // :730-755
                const double alignedFraction = double(alignment.size()) / double(n);
                if(alignedFraction < o.alignedFractionThreshold) continue;
                size_t nearDiagonalMarkerCount = 0;
                for(const auto& p : alignment) {
                    const uint32_t delta = uint32_t(std::abs(int32_t(p.first) - int32_t(p.second)));
                    if(delta < o.deltaThreshold) nearDiagonalMarkerCount++;
                }
                const double nearDiagonalFraction = double(nearDiagonalMarkerCount) / double(n);
                if(nearDiagonalFraction < o.nearDiagonalFractionThreshold) continue;
                flags[r] = uint8_t(flags[r] | 1u);
            }
        } catch(const std::exception& e) {
            std::lock_guard<std::mutex> lock(errorMutex);
            if(firstError.empty()) firstError = e.what();
        }
    };
    std::vector<std::thread> threads;
    for(size_t t = 0; t < std::min<size_t>(threadCount, std::max<size_t>(1, survivors.size())); t++) threads.emplace_back(worker);
    for(auto& t : threads) t.join();
    if(!firstError.empty()) throw std::runtime_error(firstError);

    for(uint64_t r = 0; r < readCount; r++) counts.palindromic += flags[r] & 1u;     // :690-696
    std::cout << "Flagged " << counts.palindromic << " reads as palindromic out of " << readCount << " total." << std::endl;
    std::cout << "Palindromic fraction is " << double(counts.palindromic) / double(readCount) << std::endl;
    return counts;
}

}  // namespace host
}  // namespace shasta_mi355x
