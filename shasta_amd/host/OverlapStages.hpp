// Host side of the two seams, in C++ as in the reference: the same entry points, argument lists,
// console lines and side files, over Shasta's Data/ directory -- with the work done by
// libshasta_mi355x.so through its C ABI (include/shasta_mi355x.h).
//
//   LowHash0                          <-> shasta::LowHash0::LowHash0          src/LowHash0.hpp:32-52
//   findAlignmentCandidatesLowHash0   <-> Assembler::findAlignmentCandidatesLowHash0   src/AssemblerLowHash.cpp:10-55
//   computeAlignments                 <-> Assembler::computeAlignments (alignMethod 3 or 4)  src/AssemblerAlign.cpp:208-304
//   computeAlignmentTable             <-> Assembler::computeAlignmentTable    src/AssemblerAlign.cpp:509-571
//   findMarkers                       <-> Assembler::findMarkers              src/AssemblerMarkers.cpp:11-24
//   computeCandidateTable             <-> AlignmentCandidates::computeCandidateTable    src/AssemblerAlignmentCandidates.cpp:388-447
//   createReadGraph                   <-> Assembler::createReadGraph          src/AssemblerReadGraph.cpp:35-157
//   suppressAlignmentCandidates       <-> Assembler::suppressAlignmentCandidates   src/AssemblerAlign.cpp:1168-1240
#pragma once

#include "MappedVector.hpp"
#include "../../include/shasta_mi355x.h"

#include <array>
#include <string>
#include <vector>

namespace shasta_mi355x {
namespace host {

#pragma pack(push, 1)
struct CompressedMarker7 { uint8_t bytes[7]; };          // src/Marker.hpp:56-70
#pragma pack(pop)
static_assert(sizeof(CompressedMarker7) == 7, "CompressedMarker is 7 packed bytes");

using Markers = MappedVectorOfVectors<CompressedMarker7, uint64_t>;          // Data/Markers.{toc,data}
using ReadFlagsVector = MappedVector<uint8_t>;                               // Data/ReadFlags (bit 0 = isPalindromic)
using AlignmentCandidates = MappedVector<shasta_oriented_read_pair>;         // Data/AlignmentCandidates
using ReadLowHashStatistics = MappedVector<std::array<uint64_t, 3>>;         // Data/ReadLowHashStatistics
using AlignmentDataVector = MappedVector<shasta_alignment_data>;             // Data/AlignmentData
using CompressedAlignments = MappedVectorOfVectors<char, uint64_t>;          // Data/CompressedAlignments.{toc,data}
using AlignmentTable = MappedVectorOfVectors<uint32_t, uint32_t>;            // Data/AlignmentTable.{toc,data}
using CandidateTable = MappedVectorOfVectors<uint64_t, uint64_t>;            // Data/CandidateTable.{toc,data}

// shasta::ReadGraphEdge, src/ReadGraph.hpp:37-57 (16 bytes: two OrientedReadIds, then
// alignmentId:62 | crossesStrands:1 | hasInconsistentAlignment:1 in one little-endian uint64).
struct ReadGraphEdge16 { uint32_t orientedReadIds[2]; uint64_t alignmentIdAndFlags; };
static_assert(sizeof(ReadGraphEdge16) == 16, "ReadGraphEdge is 16 bytes");
using ReadGraphEdges = MappedVector<ReadGraphEdge16>;                        // Data/ReadGraphEdges
using ReadGraphConnectivity = MappedVectorOfVectors<uint32_t, uint32_t>;     // Data/ReadGraphConnectivity.{toc,data}

// The [Align] options computeAlignments reads (src/AssemblerOptions.hpp:177-198); defaults of
// src/AssemblerOptions.cpp:380-489.
struct AlignOptions {
    int alignMethod = 4;
    uint64_t maxSkip = 30, maxDrift = 30, maxTrim = 30;
    uint64_t minAlignedMarkerCount = 100;
    double minAlignedFraction = 0.;
    int matchScore = 6, mismatchScore = -1, gapScore = -1;
    int maxBand = 1000;
    double downsamplingFactor = 0.1;          // method 3 only
    int bandExtend = 10;                      // method 3 only
    bool suppressContainments = false;
    uint64_t align4DeltaX = 200, align4DeltaY = 10, align4MinEntryCountPerCell = 10, align4MaxDistanceFromBoundary = 100;
};

// Same argument list as the reference's constructor, with the two arguments it only uses for
// the palindromic flag / not at all (Reads, kmerTable) replaced by the ReadFlags vector.  The
// constructor does all the work, like the reference's.
class LowHash0 {
public:
    LowHash0(
        size_t m, double hashFraction, size_t minHashIterationCount, double alignmentCandidatesPerRead,
        size_t log2MinHashBucketCount, size_t minBucketSize, size_t maxBucketSize, size_t minFrequency,
        size_t threadCount,
        const ReadFlagsVector& readFlags, const Markers& markers,
        AlignmentCandidates& candidates, ReadLowHashStatistics& readLowHashStatistics,
        const std::string& largeDataFileNamePrefix, size_t largeDataPageSize);
};

// dataDirectory = the run's Data/ directory (largeDataFileNamePrefix).  Side files are written to
// the current directory, as the reference does.
void findAlignmentCandidatesLowHash0(
    const std::string& dataDirectory,
    size_t m, double hashFraction, size_t minHashIterationCount, double alignmentCandidatesPerRead,
    size_t log2MinHashBucketCount, size_t minBucketSize, size_t maxBucketSize, size_t minFrequency,
    size_t threadCount, size_t largeDataPageSize = 4096);

void computeAlignments(const std::string& dataDirectory, const AlignOptions&, size_t threadCount, size_t largeDataPageSize = 4096);

// The GPUs both seams run on -- a run-level setting like the reference's --threads, not an argument of the seams (they
// keep the reference's signatures).  Default: the list in the environment variable SHASTA_MI355X_DEVICES ("0,1,2,3";
// a device may be named more than once), else device 0.  More than one device: the sharded paths of
// include/shasta_mi355x.h (*_multi), identical results.
void setDevices(const std::vector<int>& devices);
const std::vector<int>& devices();

// Assembler::findMarkers (src/AssemblerMarkers.cpp:11-24): Reads-Bases.{toc,data} + Reads-BaseCount + Kmers
// -> Markers.{toc,data}.  The step that produces the input of the two seams above.
using ReadBases = MappedVectorOfVectors<uint64_t, uint64_t>;                 // Data/Reads-Bases.{toc,data} (src/LongBaseSequence.hpp:298-301)
using ReadBaseCounts = MappedVector<uint64_t>;                               // Data/Reads-BaseCount
void findMarkers(const std::string& dataDirectory, size_t threadCount, size_t largeDataPageSize = 4096);

// AlignmentCandidates::computeCandidateTable (src/AssemblerAlignmentCandidates.cpp:388-447): the step the
// reference runs between the two seams (srcMain/main.cpp:706).  Host work: a CSR index + per-row sort.
void computeCandidateTable(uint64_t readCount, const AlignmentCandidates& candidates,
    const std::string& dataDirectory, size_t largeDataPageSize = 4096);

// Assembler::createReadGraph + createReadGraphUsingSelectedAlignments (src/AssemblerReadGraph.cpp:35-157),
// the first consumer of the alignments (ReadGraph.creationMethod 0): per read keep the
// maxAlignmentCount alignments with the most aligned markers, set AlignmentInfo::isInReadGraph in
// Data/AlignmentData, write Data/ReadGraphEdges and Data/ReadGraphConnectivity.  Returns the number kept.
uint64_t createReadGraph(const std::string& dataDirectory, uint32_t maxAlignmentCount, uint32_t maxTrim, size_t largeDataPageSize = 4096);

// Assembler::suppressAlignmentCandidates (src/AssemblerAlign.cpp:1168-1240; srcMain/main.cpp:697-702), between the two
// seams in the human Nanopore configurations: drops candidates whose reads come from the same channel / sample / run
// with `read=` numbers closer than delta.  Data/ReadNames, Data/ReadMetaData in; Data/AlignmentCandidates rewritten;
// SuppressedAlignmentCandidates.csv and the reference's three console lines.  Returns the number dropped.
uint64_t suppressAlignmentCandidates(const std::string& dataDirectory, uint64_t delta, size_t threadCount);
// ... and from keys made once per read (the meta data of a run does not change: a caller of the step parses it once): the candidates that
// stay move to the front, in order; their number is returned.  Same decisions as the string form (tests/test_candidate_suppression.py).
struct SuppressionKey { uint32_t field[3]; uint32_t flags; uint64_t read; };       // ch, sampleid, runid as numbers (0: none); flags: 1 = has a read number, 2 = it is not a number
void suppressionKeys(const uint64_t* metaDataToc, const char* metaData, uint64_t readCount, SuppressionKey* keys);
// (out: where the candidates that stay go, in order -- `candidates` itself, or an array of candidateCount entries that does not overlap it)
uint64_t suppressAlignmentCandidatesByKeys(const SuppressionKey* keys, uint64_t readCount, const shasta_oriented_read_pair* candidates, uint64_t candidateCount,
    shasta_oriented_read_pair* out, uint64_t delta, size_t threadCount);
// ... and on arrays in memory: metaDataToc[readCount + 1] offsets into metaData (the layout of Data/ReadMetaData); the candidates
// that stay are moved to the front, their number is returned.
uint64_t suppressAlignmentCandidatesInMemory(const uint64_t* metaDataToc, const char* metaData, uint64_t readCount,
    shasta_oriented_read_pair* candidates, uint64_t candidateCount, uint64_t delta);

void computeAlignmentTable(uint64_t readCount, const AlignmentDataVector& alignmentData,
    const std::string& dataDirectory, size_t largeDataPageSize = 4096);

}  // namespace host
}  // namespace shasta_mi355x
