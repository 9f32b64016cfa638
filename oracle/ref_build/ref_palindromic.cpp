// TEST INFRASTRUCTURE ONLY -- builds into oracle/_ref/libshasta_ref.so.
//
// Palindromic-read flagging on the reference's own method-0 aligner:
// /root/reference/src/AlignmentGraph.cpp is compiled in place (Makefile); Boost.Graph's three
// headers are shims/boost/graph/*.hpp.  What is restated here is the caller,
// Assembler::flagPalindromicReadsThreadFunction (src/AssemblerAlign.cpp:702-770) with
// Assembler::getMarkersSortedByKmerId (src/AssemblerMarkers.cpp:83-98): the same calls on the same
// types, so the two unstable std::sort calls and the binary-heap shortest path (src/shortestPath.hpp)
// see exactly what they see in the reference.
#include "AlignmentGraph.hpp"
#include "Alignment.hpp"
#include "Marker.hpp"
#include "ReadId.hpp"
#include "algorithm.hpp"

#include <atomic>
#include <cstring>
#include <string>
#include <thread>

using namespace shasta;

namespace { std::string palindromicError; }

extern "C" {

const char* ref_palindromic_last_error() { return palindromicError.c_str(); }

// flags[r] = 1 when read r is palindromic.  alignedCount / nearDiagonalCount (optional, per read):
// alignment.ordinals.size() and the number of aligned pairs with |ordinal0 - ordinal1| < deltaThreshold.
int ref_flag_palindromic_reads(
    uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    uint32_t maxSkip, uint32_t maxDrift, uint32_t maxMarkerFrequency,
    double alignedFractionThreshold, double nearDiagonalFractionThreshold, uint32_t deltaThreshold,
    uint64_t threadCount, uint8_t* flags, uint32_t* alignedCount, uint32_t* nearDiagonalCount, uint64_t* digests /* optional: FNV-1a of the aligned ordinals */)
{
    try {
        const CompressedMarker* all = static_cast<const CompressedMarker*>(markersData);
        if(threadCount == 0) threadCount = std::thread::hardware_concurrency();
        std::memset(flags, 0, readCount);                                   // :683-685
        std::atomic<uint64_t> next(0);
        std::string firstError;
        auto worker = [&]() {
            try {
                AlignmentGraph graph;
                Alignment alignment;
                AlignmentInfo alignmentInfo;
                array<vector<MarkerWithOrdinal>, 2> markersSortedByKmerId;
                for(;;) {
                    const uint64_t begin = next.fetch_add(1000);             // batches of 1000 reads, :688
                    if(begin >= readCount) break;
                    const uint64_t end = std::min(readCount, begin + 1000);
                    for(ReadId readId = ReadId(begin); readId != ReadId(end); readId++) {
                        for(Strand strand = 0; strand < 2; strand++) {
                            // getMarkersSortedByKmerId, src/AssemblerMarkers.cpp:83-98
                            const OrientedReadId orientedReadId(readId, strand);
                            const CompressedMarker* b = all + markersToc[orientedReadId.getValue()];
                            const CompressedMarker* e = all + markersToc[orientedReadId.getValue() + 1];
                            vector<MarkerWithOrdinal>& sorted = markersSortedByKmerId[strand];
                            sorted.clear();
                            sorted.resize(e - b);
                            for(uint32_t ordinal = 0; ordinal < uint32_t(e - b); ordinal++) {
                                sorted[ordinal] = MarkerWithOrdinal(b[ordinal], ordinal);
                            }
                            sort(sorted.begin(), sorted.end());
                        }
                        if(alignedCount) alignedCount[readId] = 0;
                        if(nearDiagonalCount) nearDiagonalCount[readId] = 0;
                        // :727-728 (Assembler::alignOrientedReads forwards to shasta::align, src/AssemblerAlign.cpp:70-86)
                        align(markersSortedByKmerId, maxSkip, maxDrift, maxMarkerFrequency, false, graph, alignment, alignmentInfo);
                        const size_t alignedMarkerCount = alignment.ordinals.size();
                        const size_t totalMarkerCount = markersSortedByKmerId[0].size();
                        size_t nearDiagonalMarkerCount = 0;
                        for(size_t i = 0; i < alignment.ordinals.size(); i++) {
                            const array<uint32_t, 2>& ordinals = alignment.ordinals[i];
                            const int32_t ordinal0 = int32_t(ordinals[0]);
                            const int32_t ordinal1 = int32_t(ordinals[1]);
                            const uint32_t delta = abs(ordinal0 - ordinal1);
                            if(delta < deltaThreshold) nearDiagonalMarkerCount++;
                        }
                        if(digests) {
                            uint64_t h = 1469598103934665603ULL;
                            for(const auto& p : alignment.ordinals) for(const uint32_t v : p) for(int b = 0; b < 4; b++) { h ^= (v >> (8 * b)) & 0xffu; h *= 1099511628211ULL; }
                            digests[readId] = h;
                        }
                        if(alignedCount) alignedCount[readId] = uint32_t(alignedMarkerCount);
                        if(nearDiagonalCount) nearDiagonalCount[readId] = uint32_t(nearDiagonalMarkerCount);
                        const double alignedFraction = double(alignedMarkerCount) / double(totalMarkerCount);
                        if(alignedFraction < alignedFractionThreshold) continue;       // :734-736
                        const double nearDiagonalFraction = double(nearDiagonalMarkerCount) / double(totalMarkerCount);
                        if(nearDiagonalFraction < nearDiagonalFractionThreshold) continue;   // :750-752
                        flags[readId] = 1;                                   // :755
                    }
                }
            } catch(std::exception& e) {
                firstError = e.what();
            }
        };
        std::vector<std::thread> threads;
        for(uint64_t t = 0; t < threadCount; t++) threads.emplace_back(worker);
        for(auto& t : threads) t.join();
        if(!firstError.empty()) throw std::runtime_error(firstError);
        return 0;
    } catch(std::exception& e) { palindromicError = e.what(); return 1; }
}

}  // extern "C"
