#include "OverlapStages.hpp"

#include <cstdlib>
#include <algorithm>
#include <fstream>
#include <functional>
#include <iostream>
#include <numeric>
#include <stdexcept>
#include <type_traits>
#include <utility>
#include <vector>

namespace shasta_mi355x {
namespace host {

namespace {
std::string dataName(const std::string& directory, const std::string& name)
{
    if(directory.empty()) return name;
    return directory.back() == '/' ? directory + name : directory + "/" + name;
}
std::vector<int>& deviceList()
{
    static std::vector<int> list = [] {
        std::vector<int> v;
        if(const char* e = std::getenv("SHASTA_MI355X_DEVICES")) {
            int value = 0; bool have = false;
            for(const char* c = e; ; c++) {
                if(*c >= '0' && *c <= '9') { value = 10 * value + (*c - '0'); have = true; }
                else { if(have) v.push_back(value); value = 0; have = false; if(*c == 0) break; }
            }
        }
        if(v.empty()) v.push_back(0);
        return v;
    }();
    return list;
}
}  // namespace

void setDevices(const std::vector<int>& d) { if(d.empty()) throw std::runtime_error("setDevices: empty device list."); deviceList() = d; }
const std::vector<int>& devices() { return deviceList(); }

LowHash0::LowHash0(
    size_t m, double hashFraction, size_t minHashIterationCount, double alignmentCandidatesPerRead,
    size_t log2MinHashBucketCount, size_t minBucketSize, size_t maxBucketSize, size_t minFrequency,
    size_t /* threadCount: the device does the work */,
    const ReadFlagsVector& readFlags, const Markers& markers,
    AlignmentCandidates& candidates, ReadLowHashStatistics& readLowHashStatistics,
    const std::string& /* largeDataFileNamePrefix: no temporary files are needed */, size_t /* largeDataPageSize */)
{
    const uint64_t orientedReadCount = markers.size();
    const uint64_t readCount = orientedReadCount / 2;
    if(readCount == 0 || readFlags.size() != readCount) throw std::runtime_error("LowHash0: markers / read flags are not consistent.");

    shasta_lowhash0_params p{};
    p.m = m; p.hashFraction = hashFraction; p.minHashIterationCount = minHashIterationCount;
    p.alignmentCandidatesPerRead = alignmentCandidatesPerRead; p.log2MinHashBucketCount = log2MinHashBucketCount;
    p.minBucketSize = minBucketSize; p.maxBucketSize = maxBucketSize; p.minFrequency = minFrequency;

    // src/LowHash0.cpp:123-125: one zeroed {sparse, good, crowded} triplet per read.
    readLowHashStatistics.resize(readCount);

    shasta_lowhash0_result r{};
    if(shasta_mi355x_lowhash0_multi(readCount, markers.toc.begin(), markers.data.begin(), readFlags.begin(), &p,
        int(devices().size()), devices().data(), reinterpret_cast<uint64_t*>(readLowHashStatistics.begin()), &r)) {
        throw std::runtime_error(shasta_mi355x_last_error());
    }

    // Console lines of src/LowHash0.cpp:89-98,193-196 ("capacity" is the reference's std::vector
    // capacity, an allocator detail: the table size is printed in its place).
    if(log2MinHashBucketCount > 31) {
        std::cout << "log2MinHashBucketCount reduced from " << log2MinHashBucketCount << " to maximum allowed value 31." << std::endl;
    }
    std::cout << "LowHash0 algorithm will use 2^" << r.log2BucketCount;
    std::cout << " = " << (1ULL << r.log2BucketCount) << " buckets. " << std::endl;
    for(uint32_t iteration = 0; iteration < r.iterationCount; iteration++) {
        // :143-146: with minHashIterationCount = 0 the reference prints the running average before every iteration but the first
        // (and once more before it stops, below).
        if(minHashIterationCount == 0 && iteration != 0) {
            std::cout << "Average number of alignment candidates that each read is involved in is " <<
                2. * double(r.highFrequency[iteration - 1]) / double(readCount) << std::endl;
        }
        std::cout << "Alignment candidates after lowhash iteration " << iteration;
        std::cout << ": high frequency " << r.highFrequency[iteration];
        std::cout << ", total " << r.total[iteration];
        std::cout << ", capacity " << r.total[iteration] << "." << std::endl;
    }
    if(minHashIterationCount == 0 && r.iterationCount != 0) {
        std::cout << "Average number of alignment candidates that each read is involved in is " <<
            2. * double(r.highFrequency[r.iterationCount - 1]) / double(readCount) << std::endl;
    }

    // LowHashBucketHistogram.csv, src/LowHash0.cpp:128,586-595.
    {
        std::ofstream csv("LowHashBucketHistogram.csv");
        csv << "Iteration,BucketSize,BucketCount,FeatureCount\n";
        for(uint64_t k = 0; k < r.histogramRowCount; k++) {
            const uint64_t iteration = r.histogram[3 * k], bucketSize = r.histogram[3 * k + 1], frequency = r.histogram[3 * k + 2];
            csv << iteration << "," << bucketSize << "," << frequency << "," << bucketSize * frequency << "\n";
        }
    }

    // src/LowHash0.cpp:204-217.
    for(uint64_t i = 0; i < r.candidateCount; i++) candidates.push_back(r.candidates[i]);
    std::cout << "Found " << candidates.size() << " alignment candidates." << std::endl;
    std::cout << "Average number of alignment candidates per oriented read is ";
    std::cout << (2. * double(candidates.size())) / double(orientedReadCount) << "." << std::endl;
    shasta_mi355x_lowhash0_free(&r);

    // ReadLowHashStatistics.csv, src/LowHash0.cpp:220-243.
    {
        std::ofstream csv("ReadLowHashStatistics.csv");
        csv << "ReadId,Palindromic,Features,Sparse,Good,Crowded,Total,FeatureSampling,"
            "SparseFraction,GoodFraction,CrowdedFraction\n";
        for(uint64_t readId = 0; readId < readCount; readId++) {
            const std::array<uint64_t, 3>& counters = readLowHashStatistics[readId];
            const uint64_t total = uint64_t(std::accumulate(counters.begin(), counters.end(), 0));   // int accumulator, as the reference
            const uint64_t featureCount = markers.size(2 * readId) - (m - 1);
            const double featureSampling = double(total) / double(featureCount);
            csv << readId << ",";
            csv << ((readFlags[readId] & 1) ? "Yes," : "No,");
            csv << featureCount << ",";
            csv << counters[0] << "," << counters[1] << "," << counters[2] << ",";
            csv << total << ",";
            csv << featureSampling << ",";
            if(total == 0) csv << ",,\n";
            else {
                csv << double(counters[0]) / double(total) << ",";
                csv << double(counters[1]) / double(total) << ",";
                csv << double(counters[2]) / double(total) << "\n";
            }
        }
    }
}

void findAlignmentCandidatesLowHash0(
    const std::string& dataDirectory,
    size_t m, double hashFraction, size_t minHashIterationCount, double alignmentCandidatesPerRead,
    size_t log2MinHashBucketCount, size_t minBucketSize, size_t maxBucketSize, size_t minFrequency,
    size_t threadCount, size_t largeDataPageSize)
{
    // checkMarkersAreOpen / reads: src/AssemblerLowHash.cpp:25-29.
    Markers markers;
    markers.accessExistingReadOnly(dataName(dataDirectory, "Markers"));
    ReadFlagsVector readFlags;
    readFlags.accessExistingReadOnly(dataName(dataDirectory, "ReadFlags"));
    if(markers.size() / 2 == 0) throw std::runtime_error("Assertion failed: readCount > 0");

    AlignmentCandidates candidates;
    ReadLowHashStatistics readLowHashStatistics;
    candidates.createNew(dataName(dataDirectory, "AlignmentCandidates"), largeDataPageSize);                 // :32
    readLowHashStatistics.createNew(dataName(dataDirectory, "ReadLowHashStatistics"), largeDataPageSize);    // :33

    LowHash0 lowHash(m, hashFraction, minHashIterationCount, alignmentCandidatesPerRead, log2MinHashBucketCount,
        minBucketSize, maxBucketSize, minFrequency, threadCount, readFlags, markers, candidates, readLowHashStatistics,
        dataDirectory, largeDataPageSize);

    candidates.unreserve();                                                                                  // :54
    readLowHashStatistics.unreserve();
}

namespace {

// Per oriented read, the indices of the pairs it takes part in, sorted by (other oriented read,
// index): the common shape of the candidate table and of the alignment table.  Built on the device
// (shasta_mi355x_pair_table: one stable radix sort of the 4 N (oriented read, partner) keys instead of the
// reference's two counting passes and a std::sort per oriented read); the sort key holds the index as
// uint32, as the reference's vector< pair<OrientedReadId, uint32_t> > does.
template<class Table>
void fillPairTable(Table& table, uint64_t readCount, const void* pairs, uint64_t strideBytes, uint64_t pairCount)
{
    using Int = typename std::remove_reference<decltype(table.toc[0])>::type;
    using Index = typename std::remove_reference<decltype(table.data[0])>::type;
    std::vector<uint64_t> toc(2 * readCount + 1, 0);
    std::vector<uint32_t> values(std::max<uint64_t>(1, 4 * pairCount));
    if(shasta_mi355x_pair_table(devices().front(), pairs, strideBytes, pairCount, readCount, toc.data(), values.data())) {
        throw std::runtime_error(shasta_mi355x_last_error());
    }
    table.toc.resize(toc.size());
    for(size_t k = 0; k < toc.size(); k++) table.toc[k] = Int(toc[k]);
    table.data.resize(4 * pairCount);
    for(uint64_t k = 0; k < 4 * pairCount; k++) table.data[k] = Index(values[k]);
    table.unreserve();
}

}  // namespace

void computeCandidateTable(uint64_t readCount, const AlignmentCandidates& candidates,
    const std::string& dataDirectory, size_t largeDataPageSize)
{
    CandidateTable table;
    table.createNew(dataName(dataDirectory, "CandidateTable"), largeDataPageSize);
    static_assert(sizeof(shasta_oriented_read_pair) == 12, "AlignmentCandidates element");
    fillPairTable(table, readCount, candidates.size() ? candidates.begin() : nullptr, sizeof(shasta_oriented_read_pair), candidates.size());
}

void computeAlignmentTable(uint64_t readCount, const AlignmentDataVector& alignmentData,
    const std::string& dataDirectory, size_t largeDataPageSize)
{
    AlignmentTable table;
    table.createNew(dataName(dataDirectory, "AlignmentTable"), largeDataPageSize);
    static_assert(sizeof(shasta_alignment_data) == 64 && offsetof(shasta_alignment_data, pair) == 0, "AlignmentData row");
    fillPairTable(table, readCount, alignmentData.size() ? alignmentData.begin() : nullptr, sizeof(shasta_alignment_data), alignmentData.size());
}

uint64_t createReadGraph(const std::string& dataDirectory, uint32_t maxAlignmentCount, uint32_t /* maxTrim: unused by the reference too */,
    size_t largeDataPageSize)
{
    AlignmentDataVector alignmentData;
    alignmentData.accessExistingReadWrite(dataName(dataDirectory, "AlignmentData"));
    AlignmentTable alignmentTable;
    alignmentTable.accessExistingReadOnly(dataName(dataDirectory, "AlignmentTable"));
    const uint64_t orientedReadCount = alignmentTable.size();
    const uint64_t readCount = orientedReadCount / 2;

    // For each read, keep only the best maxAlignmentCount alignments (:55-95): pairs (markerCount,
    // alignmentId), the largest first; ties go to the larger alignment id, as std::greater on the pair does.
    // On the device (shasta_mi355x_read_graph_keep): the reference runs an nth_element per read.
    std::vector<uint8_t> keepAlignment(alignmentData.size(), 0);
    if(shasta_mi355x_read_graph_keep(devices().front(), alignmentData.size() ? alignmentData.begin() : nullptr, alignmentData.size(), readCount,
        maxAlignmentCount, keepAlignment.data())) {
        throw std::runtime_error(shasta_mi355x_last_error());
    }
    const uint64_t keepCount = uint64_t(std::count(keepAlignment.begin(), keepAlignment.end(), uint8_t(1)));
    std::cout << "Keeping " << keepCount << " alignments of " << keepAlignment.size() << std::endl;     // :97-98

    // Edges: one per kept alignment plus its reverse complement (:115-143).
    ReadGraphEdges edges;
    edges.createNew(dataName(dataDirectory, "ReadGraphEdges"), largeDataPageSize);
    for(uint64_t alignmentId = 0; alignmentId < alignmentData.size(); alignmentId++) {
        shasta_alignment_data& alignment = alignmentData[alignmentId];
        alignment.info.isInReadGraph = keepAlignment[alignmentId] ? 1 : 0;
        if(!keepAlignment[alignmentId]) continue;
        ReadGraphEdge16 edge;
        edge.alignmentIdAndFlags = alignmentId & 0x3fffffffffffffffULL;
        edge.orientedReadIds[0] = alignment.pair.readIds[0] << 1;
        edge.orientedReadIds[1] = (alignment.pair.readIds[1] << 1) | (alignment.pair.isSameStrand ? 0u : 1u);
        edges.push_back(edge);
        edge.orientedReadIds[0] ^= 1u; edge.orientedReadIds[1] ^= 1u;
        edges.push_back(edge);
    }
    edges.unreserve();

    // Connectivity: per oriented read the indices of its edges.  The reference fills each row from
    // the back (VectorOfVectors::store, src/MemoryMappedVectorOfVectors.hpp:383-386): descending edge index.
    ReadGraphConnectivity connectivity;
    connectivity.createNew(dataName(dataDirectory, "ReadGraphConnectivity"), largeDataPageSize);
    std::vector<uint32_t> counts(orientedReadCount, 0);
    for(uint64_t i = 0; i < edges.size(); i++) { ++counts[edges[i].orientedReadIds[0]]; ++counts[edges[i].orientedReadIds[1]]; }
    connectivity.fillFromCounts(counts);
    for(uint64_t i = 0; i < edges.size(); i++) {
        for(int k = 0; k < 2; k++) {
            const uint32_t o = edges[i].orientedReadIds[k];
            connectivity.begin(o)[--counts[o]] = uint32_t(i);
        }
    }
    connectivity.unreserve();

    uint64_t isolatedReadCount = 0;
    for(uint64_t readId = 0; readId < readCount; readId++) if(connectivity.size(2 * readId) == 0) ++isolatedReadCount;
    std::cout << "The read graph has " << edges.size() / 2 << " edges (and as many reverse complemented); "
        << isolatedReadCount << " reads are isolated." << std::endl;
    return keepCount;
}

void findMarkers(const std::string& dataDirectory, size_t /* threadCount */, size_t largeDataPageSize)
{
    struct KmerInfo24 { char bytes[24]; };                 // src/Kmer.hpp:22-39: isMarker is byte 12
    ReadBases bases;
    bases.accessExistingReadOnly(dataName(dataDirectory, "Reads-Bases"));
    ReadBaseCounts baseCounts;
    baseCounts.accessExistingReadOnly(dataName(dataDirectory, "Reads-BaseCount"));
    MappedVector<KmerInfo24> kmers;
    kmers.accessExistingReadOnly(dataName(dataDirectory, "Kmers"));
    const uint64_t readCount = baseCounts.size();
    if(bases.size() != readCount) throw std::runtime_error("findMarkers: Reads-Bases and Reads-BaseCount disagree.");
    uint64_t k = 0;
    while(k < 32 && (1ULL << (2 * k)) < kmers.size()) ++k;
    if(kmers.size() == 0 || (1ULL << (2 * k)) != kmers.size()) throw std::runtime_error("findMarkers: Data/Kmers does not hold 4^k entries.");

    shasta_markers_result r{};
    if(shasta_mi355x_find_markers(nullptr, readCount, bases.toc.begin(), bases.data.begin(), baseCounts.begin(),
        k, kmers.begin(), sizeof(KmerInfo24), 12, nullptr, 1, &r)) {
        throw std::runtime_error(shasta_mi355x_last_error());
    }
    Markers markers;
    markers.createNew(dataName(dataDirectory, "Markers"), largeDataPageSize);                                // :16
    for(uint64_t i = 0; i < 2 * readCount; i++) {
        markers.appendVector(reinterpret_cast<const CompressedMarker7*>(r.markersData + 7 * r.markersToc[i]),
            r.markersToc[i + 1] - r.markersToc[i]);
    }
    shasta_mi355x_find_markers_free(&r);
    markers.unreserve();                                                                                     // MarkerFinder.cpp:49
}

// The marker length k of the run, from the size of Data/Kmers (Vector<KmerInfo>, 24-byte entries,
// one per k-mer id: 4^k of them, src/AssemblerKmers.cpp:147-186).  Method 3 needs it for the
// down-sampling hash, which the library recomputes from the k-mer id instead of reading the table.
uint64_t markerLengthOf(const std::string& dataDirectory)
{
    struct KmerInfo24 { char bytes[24]; };                 // src/Kmer.hpp:22-39
    MappedVector<KmerInfo24> kmers;
    kmers.accessExistingReadOnly(dataName(dataDirectory, "Kmers"));
    const uint64_t n = kmers.size();
    uint64_t k = 0;
    while(k < 32 && (1ULL << (2 * k)) < n) ++k;
    if(n == 0 || (1ULL << (2 * k)) != n) throw std::runtime_error("computeAlignments: Data/Kmers does not hold 4^k entries.");
    return k;
}

void computeAlignments(const std::string& dataDirectory, const AlignOptions& alignOptions, size_t /* threadCount */, size_t largeDataPageSize)
{
    if(alignOptions.alignMethod != 3 && alignOptions.alignMethod != 4) {
        throw std::runtime_error("computeAlignments: this library implements alignMethod 3 and 4 only.");
    }
    Markers markers;
    markers.accessExistingReadOnly(dataName(dataDirectory, "Markers"));
    AlignmentCandidates candidates;
    candidates.accessExistingReadOnly(dataName(dataDirectory, "AlignmentCandidates"));
    const uint64_t readCount = markers.size() / 2;

    shasta_align4_result r{};
    if(alignOptions.alignMethod == 3) {
        // src/AssemblerAlign.cpp:404-409 -> Assembler::alignOrientedReads3.
        shasta_align3_options o{};
        o.matchScore = alignOptions.matchScore; o.mismatchScore = alignOptions.mismatchScore; o.gapScore = alignOptions.gapScore;
        o.downsamplingFactor = alignOptions.downsamplingFactor;
        o.bandExtend = alignOptions.bandExtend; o.maxBand = alignOptions.maxBand;
        o.k = markerLengthOf(dataDirectory);
        o.minAlignedMarkerCount = alignOptions.minAlignedMarkerCount;
        o.minAlignedFraction = alignOptions.minAlignedFraction;
        o.maxSkip = alignOptions.maxSkip; o.maxDrift = alignOptions.maxDrift; o.maxTrim = alignOptions.maxTrim;
        o.suppressContainments = alignOptions.suppressContainments ? 1 : 0;
        if(shasta_mi355x_align3_batch_multi(readCount, markers.toc.begin(), markers.data.begin(),
            candidates.size(), candidates.begin(), &o, 0, int(devices().size()), devices().data(), &r)) {
            throw std::runtime_error(shasta_mi355x_last_error());
        }
    } else {
        shasta_align4_options o{};
        o.deltaX = alignOptions.align4DeltaX; o.deltaY = alignOptions.align4DeltaY;
        o.minEntryCountPerCell = alignOptions.align4MinEntryCountPerCell;
        o.maxDistanceFromBoundary = alignOptions.align4MaxDistanceFromBoundary;
        o.minAlignedMarkerCount = alignOptions.minAlignedMarkerCount;
        o.minAlignedFraction = alignOptions.minAlignedFraction;
        o.maxSkip = alignOptions.maxSkip; o.maxDrift = alignOptions.maxDrift; o.maxTrim = alignOptions.maxTrim;
        o.maxBand = uint64_t(alignOptions.maxBand);
        o.matchScore = alignOptions.matchScore; o.mismatchScore = alignOptions.mismatchScore; o.gapScore = alignOptions.gapScore;
        o.suppressContainments = alignOptions.suppressContainments ? 1 : 0;
        if(shasta_mi355x_align4_batch_multi(readCount, markers.toc.begin(), markers.data.begin(),
            candidates.size(), candidates.begin(), &o, 0, int(devices().size()), devices().data(), &r)) {
            throw std::runtime_error(shasta_mi355x_last_error());
        }
    }
    uint64_t skipped = 0;
    for(uint64_t i = 0; i < candidates.size(); i++) if((r.status[i] & 0x7f) == SHASTA_ALIGN_SKIPPED) ++skipped;
    if(skipped) std::cout << skipped << " alignment candidates were skipped (resource limits), as the reference skips candidates whose alignment throws." << std::endl;

    AlignmentDataVector alignmentData;
    CompressedAlignments compressedAlignments;
    alignmentData.createNew(dataName(dataDirectory, "AlignmentData"), largeDataPageSize);                    // :263
    compressedAlignments.createNew(dataName(dataDirectory, "CompressedAlignments"), largeDataPageSize);      // :264
    alignmentData.append(r.alignmentData, r.alignmentCount);
    for(uint64_t i = 0; i < r.alignmentCount; i++) {
        compressedAlignments.appendVector(reinterpret_cast<const char*>(r.compressedData + r.compressedToc[i]),
            r.compressedToc[i + 1] - r.compressedToc[i]);
    }
    shasta_mi355x_align4_free(&r);
    alignmentData.unreserve();                                                                               // :286-287
    compressedAlignments.unreserve();
    std::cout << "Found and stored " << alignmentData.size() << " good alignments." << std::endl;           // :294
    computeAlignmentTable(readCount, alignmentData, dataDirectory, largeDataPageSize);                        // :296
}

}  // namespace host
}  // namespace shasta_mi355x
