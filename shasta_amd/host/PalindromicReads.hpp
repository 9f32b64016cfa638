// Assembler::flagPalindromicReads (src/AssemblerAlign.cpp:652-770) over a run's Data/ directory.
//
// Two halves.  The device screens every read: shasta_mi355x_palindromic_screen gives an upper bound on
// the near-diagonal marker count of the read's self-alignment, and a read whose bound is below the
// threshold cannot be flagged, whatever the alignment is (exact).  The host decides the few that
// remain with the reference's own procedure -- alignment method 0: markers sorted by kmer id, the
// alignment graph, the lazy-deletion shortest path (src/AlignmentGraph.cpp:58-133,
// src/shortestPath.hpp:57-161).  That procedure is sequential and its result depends on the order in
// which std::sort leaves equal keys and std::priority_queue pops equal distances, so it is kept as
// those very library calls on the host: same calls on the same key sequences, same answer.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace shasta_mi355x {
namespace host {

// [Reads.palindromicReads] options, defaults of src/AssemblerOptions.cpp:255-288.
struct PalindromicReadOptions {
    uint32_t maxSkip = 100, maxDrift = 100, maxMarkerFrequency = 10;
    double alignedFractionThreshold = 0.1, nearDiagonalFractionThreshold = 0.1;
    uint32_t deltaThreshold = 100;
};

// Alignment method 0 of one read (strand 0) against its reverse complement (strand 1): the aligned
// ordinal pairs in path order.  kmerIds0 / kmerIds1: the n marker kmer ids of the two strands.
void selfAlignmentMethod0(const uint32_t* kmerIds0, const uint32_t* kmerIds1, uint32_t n,
    uint32_t maxSkip, uint32_t maxDrift, uint32_t maxMarkerFrequency, std::vector<std::pair<uint32_t, uint32_t>>& alignment);

struct PalindromicReadCounts { uint64_t readCount = 0, screenedOut = 0, palindromic = 0; };

// Sets / clears bit 0 (isPalindromic, src/ReadFlags.hpp:10-30) of every read's flag byte in
// Data/ReadFlags and prints the reference's two console lines (:694-697).
PalindromicReadCounts flagPalindromicReads(const std::string& dataDirectory, const PalindromicReadOptions&, size_t threadCount);

}  // namespace host
}  // namespace shasta_mi355x
