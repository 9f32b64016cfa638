// C entry points over the host layer's file classes, for the Python tests only
// (tests/test_host_files.py); not part of the product ABI.
#include "OverlapStages.hpp"

#include <cstring>
#include <string>

using namespace shasta_mi355x::host;

static thread_local std::string shimError;
#define SHIM_BEGIN try {
#define SHIM_END } catch(const std::exception& e) { shimError = e.what(); return 1; } return 0;

namespace {
template<size_t N> struct Blob { char bytes[N]; };
template<size_t N> void readBlobVector(const char* path, uint64_t* count, uint64_t* fileBytes, void* out, uint64_t capacity)
{
    MappedVector< Blob<N> > v;
    v.accessExistingReadOnly(path);
    *count = v.size();
    *fileBytes = 4096 + N * v.capacity();
    if(out) {
        if(N * v.size() > capacity) throw std::runtime_error("output capacity too small");
        if(v.size()) std::memcpy(out, v.begin(), N * v.size());
    }
}
}  // namespace

extern "C" {

const char* host_last_error() { return shimError.c_str(); }

int host_write_data_dir(const char* dir, uint64_t readCount, const uint64_t* toc, const void* data7, const uint8_t* flags)
{
    SHIM_BEGIN
    const std::string d(dir);
    Markers markers;
    markers.createNew(d + "/Markers");
    for(uint64_t i = 0; i < 2 * readCount; i++) {
        markers.appendVector(reinterpret_cast<const CompressedMarker7*>(static_cast<const char*>(data7) + 7 * toc[i]), toc[i + 1] - toc[i]);
    }
    markers.unreserve();
    ReadFlagsVector readFlags;
    readFlags.createNew(d + "/ReadFlags");
    readFlags.resize(readCount);
    for(uint64_t i = 0; i < readCount; i++) readFlags[i] = flags ? flags[i] : 0;
    readFlags.unreserve();
    SHIM_END
}

// Data/ReadFlags alone (the reference's ReadLoader writes it with the reads).
int host_write_read_flags(const char* dir, uint64_t readCount, const uint8_t* flags)
{
    SHIM_BEGIN
    ReadFlagsVector readFlags;
    readFlags.createNew(std::string(dir) + "/ReadFlags");
    readFlags.resize(readCount);
    for(uint64_t i = 0; i < readCount; i++) readFlags[i] = flags ? flags[i] : 0;
    readFlags.unreserve();
    SHIM_END
}

int host_open_vector(const char* path, uint64_t objectSize, uint64_t* objectCount, uint64_t* fileBytes, void* out, uint64_t capacity)
{
    SHIM_BEGIN
    switch(objectSize) {
        case 1: readBlobVector<1>(path, objectCount, fileBytes, out, capacity); break;
        case 4: readBlobVector<4>(path, objectCount, fileBytes, out, capacity); break;
        case 7: readBlobVector<7>(path, objectCount, fileBytes, out, capacity); break;
        case 8: readBlobVector<8>(path, objectCount, fileBytes, out, capacity); break;
        case 12: readBlobVector<12>(path, objectCount, fileBytes, out, capacity); break;
        case 16: readBlobVector<16>(path, objectCount, fileBytes, out, capacity); break;
        case 24: readBlobVector<24>(path, objectCount, fileBytes, out, capacity); break;
        case 64: readBlobVector<64>(path, objectCount, fileBytes, out, capacity); break;
        default: throw std::runtime_error("unsupported object size");
    }
    SHIM_END
}

int host_store_alignments(const char* dir, uint64_t alignmentCount, const shasta_alignment_data* rows,
    const uint64_t* compressedToc, const uint8_t* compressedData)
{
    SHIM_BEGIN
    const std::string d(dir);
    AlignmentDataVector alignmentData;
    CompressedAlignments compressed;
    alignmentData.createNew(d + "/AlignmentData");
    compressed.createNew(d + "/CompressedAlignments");
    alignmentData.append(rows, alignmentCount);
    for(uint64_t i = 0; i < alignmentCount; i++) {
        compressed.appendVector(reinterpret_cast<const char*>(compressedData + compressedToc[i]), compressedToc[i + 1] - compressedToc[i]);
    }
    alignmentData.unreserve();
    compressed.unreserve();
    SHIM_END
}

// Data/Kmers with the 4^k entries of a run with marker length k: zero except the isMarker byte
// (offset 12 of KmerInfo, src/Kmer.hpp:22-39) when isMarker is given.
int host_write_kmers(const char* dir, uint64_t k, const uint8_t* isMarker)
{
    SHIM_BEGIN
    MappedVector< Blob<24> > kmers;
    kmers.createNew(std::string(dir) + "/Kmers");
    kmers.resize(1ULL << (2 * k));
    std::memset(kmers.begin(), 0, 24 * kmers.size());
    if(isMarker) for(uint64_t i = 0; i < kmers.size(); i++) kmers[i].bytes[12] = char(isMarker[i] ? 1 : 0);
    kmers.unreserve();
    SHIM_END
}

// Data/Reads-Bases.{toc,data} and Data/Reads-BaseCount (LongBaseSequences, src/LongBaseSequence.cpp:8-20).
int host_write_reads(const char* dir, uint64_t readCount, const uint64_t* readsToc, const uint64_t* readsData, const uint64_t* baseCounts)
{
    SHIM_BEGIN
    const std::string d(dir);
    ReadBases bases;
    bases.createNew(d + "/Reads-Bases");
    for(uint64_t r = 0; r < readCount; r++) bases.appendVector(readsData + readsToc[r], readsToc[r + 1] - readsToc[r]);
    bases.unreserve();
    ReadBaseCounts counts;
    counts.createNew(d + "/Reads-BaseCount");
    counts.append(baseCounts, readCount);
    counts.unreserve();
    SHIM_END
}

int host_store_candidates(const char* dir, uint64_t count, const shasta_oriented_read_pair* pairs)
{
    SHIM_BEGIN
    AlignmentCandidates candidates;
    candidates.createNew(std::string(dir) + "/AlignmentCandidates");
    candidates.append(pairs, count);
    candidates.unreserve();
    SHIM_END
}

int host_compute_candidate_table(const char* dir, uint64_t readCount)
{
    SHIM_BEGIN
    const std::string d(dir);
    AlignmentCandidates candidates;
    candidates.accessExistingReadOnly(d + "/AlignmentCandidates");
    computeCandidateTable(readCount, candidates, d);
    SHIM_END
}

int host_create_read_graph(const char* dir, uint32_t maxAlignmentCount, uint64_t* keepCount)
{
    SHIM_BEGIN
    *keepCount = createReadGraph(dir, maxAlignmentCount, 30);
    SHIM_END
}

int host_compute_alignment_table(const char* dir, uint64_t readCount)
{
    SHIM_BEGIN
    const std::string d(dir);
    AlignmentDataVector alignmentData;
    alignmentData.accessExistingReadOnly(d + "/AlignmentData");
    computeAlignmentTable(readCount, alignmentData, d);
    SHIM_END
}

}  // extern "C"
