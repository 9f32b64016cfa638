// C entry points of the host layer (libshasta_mi355x_host.so): the stage functions of
// OverlapStages.hpp over a Data/ directory, for callers that are not C++ (the Python
// shasta_amd.Assembler mirror of the reference's pybind11 class, src/PythonModule.cpp:135-345).
// Returns 0 on success; shasta_mi355x_host_last_error() gives the message otherwise.
#include "OverlapStages.hpp"
#include "PalindromicReads.hpp"

#include <string>

using namespace shasta_mi355x::host;

static thread_local std::string hostError;
#define HOST_BEGIN try {
#define HOST_END } catch(const std::exception& e) { hostError = e.what(); return 1; } return 0;

extern "C" {

const char* shasta_mi355x_host_last_error(void) { return hostError.c_str(); }

// Assembler::findAlignmentCandidatesLowHash0, src/AssemblerLowHash.cpp:10-55.
int shasta_mi355x_host_find_alignment_candidates_lowhash0(
    const char* dataDirectory, uint64_t m, double hashFraction, uint64_t minHashIterationCount,
    double alignmentCandidatesPerRead, uint64_t log2MinHashBucketCount, uint64_t minBucketSize,
    uint64_t maxBucketSize, uint64_t minFrequency, uint64_t threadCount, uint64_t largeDataPageSize)
{
    HOST_BEGIN
    findAlignmentCandidatesLowHash0(dataDirectory, m, hashFraction, minHashIterationCount, alignmentCandidatesPerRead,
        log2MinHashBucketCount, minBucketSize, maxBucketSize, minFrequency, threadCount, largeDataPageSize);
    HOST_END
}

// Assembler::findMarkers, src/AssemblerMarkers.cpp:11-24.
int shasta_mi355x_host_find_markers(const char* dataDirectory, uint64_t threadCount, uint64_t largeDataPageSize)
{
    HOST_BEGIN
    findMarkers(dataDirectory, threadCount, largeDataPageSize);
    HOST_END
}

// Assembler::computeCandidateTable, src/AssemblerAlignmentCandidates.cpp:379-447.
int shasta_mi355x_host_compute_candidate_table(const char* dataDirectory, uint64_t largeDataPageSize)
{
    HOST_BEGIN
    const std::string d(dataDirectory);
    Markers markers;
    markers.accessExistingReadOnly(d + "/Markers");
    AlignmentCandidates candidates;
    candidates.accessExistingReadOnly(d + "/AlignmentCandidates");
    computeCandidateTable(markers.size() / 2, candidates, d, largeDataPageSize);
    HOST_END
}

// The numeric members of AlignOptions that methods 3 and 4 read, in the order of src/AssemblerOptions.hpp:177-198.
struct shasta_mi355x_host_align_options {
    int64_t alignMethod;
    uint64_t maxSkip, maxDrift, maxTrim, minAlignedMarkerCount;
    double minAlignedFraction;
    int64_t matchScore, mismatchScore, gapScore, maxBand;
    uint64_t suppressContainments;
    uint64_t align4DeltaX, align4DeltaY, align4MinEntryCountPerCell, align4MaxDistanceFromBoundary;
    double downsamplingFactor;     /* method 3 */
    int64_t bandExtend;            /* method 3 */
};

// Assembler::computeAlignments, src/AssemblerAlign.cpp:208-304.
int shasta_mi355x_host_compute_alignments(const char* dataDirectory, const shasta_mi355x_host_align_options* o,
    uint64_t threadCount, uint64_t largeDataPageSize)
{
    HOST_BEGIN
    AlignOptions a;
    a.alignMethod = int(o->alignMethod);
    a.maxSkip = o->maxSkip; a.maxDrift = o->maxDrift; a.maxTrim = o->maxTrim;
    a.minAlignedMarkerCount = o->minAlignedMarkerCount; a.minAlignedFraction = o->minAlignedFraction;
    a.matchScore = int(o->matchScore); a.mismatchScore = int(o->mismatchScore); a.gapScore = int(o->gapScore);
    a.maxBand = int(o->maxBand); a.suppressContainments = o->suppressContainments != 0;
    a.align4DeltaX = o->align4DeltaX; a.align4DeltaY = o->align4DeltaY;
    a.align4MinEntryCountPerCell = o->align4MinEntryCountPerCell;
    a.align4MaxDistanceFromBoundary = o->align4MaxDistanceFromBoundary;
    a.downsamplingFactor = o->downsamplingFactor; a.bandExtend = int(o->bandExtend);
    computeAlignments(dataDirectory, a, threadCount, largeDataPageSize);
    HOST_END
}

// Assembler::createReadGraph, src/AssemblerReadGraph.cpp:35-104.
int shasta_mi355x_host_create_read_graph(const char* dataDirectory, uint32_t maxAlignmentCount, uint32_t maxTrim, uint64_t largeDataPageSize)
{
    HOST_BEGIN
    (void)createReadGraph(dataDirectory, maxAlignmentCount, maxTrim, largeDataPageSize);
    HOST_END
}

// Assembler::suppressAlignmentCandidates, src/AssemblerAlign.cpp:1168-1240.
int shasta_mi355x_host_suppress_alignment_candidates(const char* dataDirectory, uint64_t delta, uint64_t threadCount, uint64_t* suppressed)
{
    HOST_BEGIN
    const uint64_t n = suppressAlignmentCandidates(dataDirectory, delta, threadCount);
    if(suppressed) *suppressed = n;
    HOST_END
}

// The same decision on arrays in memory: the candidates that stay move to the front, *kept = their number.
int shasta_mi355x_host_suppress_candidates_in_memory(const uint64_t* metaDataToc, const char* metaData, uint64_t readCount,
    shasta_oriented_read_pair* candidates, uint64_t candidateCount, uint64_t delta, uint64_t* kept)
{
    HOST_BEGIN
    if(!metaDataToc || !metaData || (!candidates && candidateCount) || !kept) throw std::runtime_error("suppress_candidates_in_memory: null argument");
    *kept = suppressAlignmentCandidatesInMemory(metaDataToc, metaData, readCount, candidates, candidateCount, delta);
    HOST_END
}

// ... from keys made once per read: keys = 24 bytes per read (SuppressionKey), filled by the first call, read by the second.
int shasta_mi355x_host_suppression_keys(const uint64_t* metaDataToc, const char* metaData, uint64_t readCount, void* keys)
{
    HOST_BEGIN
    if(!metaDataToc || !metaData || !keys) throw std::runtime_error("suppression_keys: null argument");
    suppressionKeys(metaDataToc, metaData, readCount, static_cast<SuppressionKey*>(keys));
    HOST_END
}
int shasta_mi355x_host_suppress_candidates_by_keys(const void* keys, uint64_t readCount, const shasta_oriented_read_pair* candidates, uint64_t candidateCount,
    shasta_oriented_read_pair* out, uint64_t delta, uint64_t threadCount, uint64_t* kept)
{
    HOST_BEGIN
    if(!keys || ((!candidates || !out) && candidateCount) || !kept) throw std::runtime_error("suppress_candidates_by_keys: null argument");
    *kept = suppressAlignmentCandidatesByKeys(static_cast<const SuppressionKey*>(keys), readCount, candidates, candidateCount, out, delta, size_t(threadCount));
    HOST_END
}

// Assembler::flagPalindromicReads, src/AssemblerAlign.cpp:652-698.  alignment (optional): for tests, the method-0
// self-alignment of one read is available through shasta_mi355x_host_self_alignment_method0.
int shasta_mi355x_host_flag_palindromic_reads(const char* dataDirectory, uint32_t maxSkip, uint32_t maxDrift, uint32_t maxMarkerFrequency,
    double alignedFractionThreshold, double nearDiagonalFractionThreshold, uint32_t deltaThreshold, uint64_t threadCount,
    uint64_t* counts /* optional: readCount, screenedOut, palindromic */)
{
    HOST_BEGIN
    PalindromicReadOptions o;
    o.maxSkip = maxSkip; o.maxDrift = maxDrift; o.maxMarkerFrequency = maxMarkerFrequency;
    o.alignedFractionThreshold = alignedFractionThreshold; o.nearDiagonalFractionThreshold = nearDiagonalFractionThreshold;
    o.deltaThreshold = deltaThreshold;
    const PalindromicReadCounts c = flagPalindromicReads(dataDirectory, o, threadCount);
    if(counts) { counts[0] = c.readCount; counts[1] = c.screenedOut; counts[2] = c.palindromic; }
    HOST_END
}

// Alignment method 0 of a read against its reverse complement (the host half of the step above), for unit
// parity: ordinals gets 2 * (*count) values, (ordinal0, ordinal1) per aligned marker; capacity is in pairs.
int shasta_mi355x_host_self_alignment_method0(const uint32_t* kmerIds0, const uint32_t* kmerIds1, uint32_t n,
    uint32_t maxSkip, uint32_t maxDrift, uint32_t maxMarkerFrequency, uint32_t* ordinals, uint64_t capacity, uint64_t* count)
{
    HOST_BEGIN
    std::vector<std::pair<uint32_t, uint32_t>> alignment;
    selfAlignmentMethod0(kmerIds0, kmerIds1, n, maxSkip, maxDrift, maxMarkerFrequency, alignment);
    if(alignment.size() > capacity) throw std::runtime_error("self_alignment_method0: output capacity too small.");
    for(size_t i = 0; i < alignment.size(); i++) { ordinals[2 * i] = alignment[i].first; ordinals[2 * i + 1] = alignment[i].second; }
    *count = alignment.size();
    HOST_END
}

}  // extern "C"
