// TEST INFRASTRUCTURE ONLY -- builds into oracle/_ref/libshasta_ref.so.
//
// C-ABI harness around the REFERENCE's own translation units, compiled in place
// from /root/reference/src (see Makefile; no reference source is copied into
// this repository).  It drives, without the Assembler facade (which would pull
// in Boost):
//
//   ReadLoader -> (restated randomlySelectKmers) -> MarkerFinder      [fixture generation]
//   LowHash0::LowHash0                                                [seam 1 oracle]
//   Align4::align + AlignmentInfo + shasta::compress                  [seam 2 oracle]
//   Assembler::alignOrientedReads3 (ref_align3.cpp / ref_facade.hpp)  [align method 3 oracle]
//
// Non-reference code in this file: the k-mer table fill (restating
// src/AssemblerKmers.cpp:33-100,147-180) and the per-candidate driver loop
// (restating src/AssemblerAlign.cpp:378-483 for alignMethod 4, threadCount 1).
// The banded DP behind Align4 is NOT SeqAn: see shims/seqan/align.h.

// Reads keeps its vectors private; the harness needs to size readFlags without
// loading bases.  Access control does not change layout.
// (standard headers first, so that only Shasta headers see the redefinition)
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <atomic>
#include <map>
#include <mutex>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>
#define private public
#include "Reads.hpp"
#undef private

#include "Align4.hpp"
#include "Alignment.hpp"
#include "compressAlignment.hpp"
#include "Kmer.hpp"
#include "LowHash0.hpp"
#include "Marker.hpp"
#include "MarkerFinder.hpp"
#include "MemoryMappedAllocator.hpp"
#include "MemoryMappedVector.hpp"
#include "MemoryMappedVectorOfVectors.hpp"
#include "MurmurHash2.hpp"
#include "OrientedReadPair.hpp"
#include "orderPairs.hpp"
#include "ReadLoader.hpp"
#include "ref_facade.hpp"
using namespace shasta;

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <unistd.h>

#include "../../include/shasta_mi355x.h"
#include "../banded_dp.hpp"

static_assert(sizeof(CompressedMarker) == 7, "CompressedMarker");
static_assert(sizeof(OrientedReadPair) == sizeof(shasta_oriented_read_pair), "OrientedReadPair");
static_assert(sizeof(AlignmentInfo) == sizeof(shasta_alignment_info), "AlignmentInfo");
static_assert(sizeof(AlignmentData) == sizeof(shasta_alignment_data), "AlignmentData");
static_assert(sizeof(Align4::Options) == 104, "Align4::Options");

static thread_local std::string lastError;

namespace {

using Markers = MemoryMapped::VectorOfVectors<CompressedMarker, uint64_t>;

void fillMarkers(Markers& markers, uint64_t readCount, const uint64_t* toc, const void* data)
{
    markers.createNew("", 4096);
    const uint64_t n = 2 * readCount;
    markers.beginPass1(n);
    for(uint64_t i = 0; i < n; i++) {
        markers.incrementCount(i, toc[i+1] - toc[i]);
    }
    markers.beginPass2();
    markers.endPass2(false);
    if(toc[n]) {
        std::memcpy(markers.begin(), data, 7ULL * toc[n]);
    }
}

class CoutCapture {
public:
    CoutCapture() : old(std::cout.rdbuf(ss.rdbuf())) {}
    ~CoutCapture() { std::cout.rdbuf(old); }
    std::string str() const { return ss.str(); }
private:
    std::stringstream ss;
    std::streambuf* old;
};

class ChdirGuard {
public:
    explicit ChdirGuard(const char* dir)
    {
        if(!getcwd(old, sizeof(old))) old[0] = 0;
        if(dir && dir[0]) { if(chdir(dir)) throw std::runtime_error("chdir failed"); }
    }
    ~ChdirGuard() { if(old[0]) { if(chdir(old)) {} } }
private:
    char old[4096];
};

template<class T> T* mallocCopy(const std::vector<T>& v)
{
    T* p = static_cast<T*>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
    if(!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

}  // namespace


extern "C" {

const char* ref_last_error() { return lastError.c_str(); }

uint64_t ref_murmur64a(const void* key, int len, uint64_t seed)
{
    return MurmurHash64A(key, len, seed);
}

// Calls the reference's own codec self test (src/compressAlignment.cpp:160-220).
int ref_test_alignment_compression()
{
    try { CoutCapture c; testAlignmentCompression(); return 0; }
    catch(std::exception& e) { lastError = e.what(); return 1; }
}

// shasta::compress on an ordinal list; returns malloc'd bytes.
int ref_compress(const uint32_t* ordinals, uint64_t n, uint8_t** bytes, uint64_t* byteCount)
{
    try {
        Alignment a;
        a.ordinals.resize(n);
        for(uint64_t i = 0; i < n; i++) a.ordinals[i] = {ordinals[2*i], ordinals[2*i+1]};
        string s;
        compress(a, s);
        *byteCount = s.size();
        *bytes = static_cast<uint8_t*>(std::malloc(std::max<size_t>(1, s.size())));
        std::memcpy(*bytes, s.data(), s.size());
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

int ref_decompress(const uint8_t* bytes, uint64_t byteCount, uint32_t** ordinals, uint64_t* n)
{
    try {
        Alignment a;
        const char* p = reinterpret_cast<const char*>(bytes);
        decompress(span<const char>(p, p + byteCount), a);
        *n = a.ordinals.size();
        *ordinals = static_cast<uint32_t*>(std::malloc(std::max<size_t>(1, 8 * a.ordinals.size())));
        for(uint64_t i = 0; i < a.ordinals.size(); i++) {
            (*ordinals)[2*i] = a.ordinals[i][0];
            (*ordinals)[2*i+1] = a.ordinals[i][1];
        }
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

// AlignmentInfo::create (src/Alignment.cpp:67-113).
int ref_alignment_info(const uint32_t* ordinals, uint64_t n, uint32_t nx, uint32_t ny,
    shasta_alignment_info* out)
{
    try {
        Alignment a;
        a.ordinals.resize(n);
        for(uint64_t i = 0; i < n; i++) a.ordinals[i] = {ordinals[2*i], ordinals[2*i+1]};
        AlignmentInfo info;
        info.create(a, nx, ny);
        std::memset(out, 0, sizeof(*out));
        for(int i = 0; i < 2; i++) {
            out->data[i].markerCount = info.data[i].markerCount;
            out->data[i].firstOrdinal = info.data[i].firstOrdinal;
            out->data[i].lastOrdinal = info.data[i].lastOrdinal;
        }
        out->markerCount = info.markerCount;
        out->minOrdinalOffset = info.minOrdinalOffset;
        out->maxOrdinalOffset = info.maxOrdinalOffset;
        out->averageOrdinalOffset = info.averageOrdinalOffset;
        out->maxSkip = info.maxSkip;
        out->maxDrift = info.maxDrift;
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

void ref_free(void* p) { std::free(p); }


// --------------------------------------------------------------------------
// FASTA -> markers, with the reference's ReadLoader and MarkerFinder.
// --------------------------------------------------------------------------
int ref_markers_from_fasta(
    const char* fastaPath, uint64_t k, double probability, int seed,
    uint64_t minReadLength, uint64_t threadCount,
    uint64_t* readCountOut, uint64_t** tocOut, uint8_t** dataOut)
{
    try {
        CoutCapture capture;
        if(threadCount == 0) threadCount = std::thread::hardware_concurrency();
        Reads reads;
        reads.createNew(1, "", "", "", "", "", "", 4096);
        {
            ReadLoader loader(fastaPath, 1, minReadLength, false, threadCount, "", 4096, reads);
        }

        // K-mer table: src/AssemblerKmers.cpp:147-180 (initializeKmerTable) and
        // :33-100 (randomlySelectKmers), restated.
        MemoryMapped::Vector<KmerInfo> kmerTable;
        kmerTable.createNew("", 4096);
        const uint64_t kmerCount = 1ULL << (2ULL * k);
        kmerTable.resize(kmerCount);
        for(uint64_t kmerId = 0; kmerId < kmerCount; kmerId++) {
            const Kmer kmer(kmerId, k);
            KmerInfo& info = kmerTable[kmerId];
            info.frequency = 0;
            info.reverseComplementedKmerId = KmerId(kmer.reverseComplement(k).id(k));
            info.isMarker = false;
            info.isRleKmer = true;
            for(size_t i = 1; i < k; i++) {
                if(kmer[i-1] == kmer[i]) { info.isRleKmer = false; break; }
            }
            info.hash = 0;
        }
        const double p = 1. - std::sqrt(1. - probability);
        std::mt19937 randomSource(seed);
        std::uniform_real_distribution<> uniformDistribution;
        for(uint64_t kmerId = 0; kmerId < kmerCount; kmerId++) {
            const double x = uniformDistribution(randomSource);
            if(x <= p) {
                kmerTable[kmerId].isMarker = true;
                kmerTable[kmerTable[kmerId].reverseComplementedKmerId].isMarker = true;
            }
        }

        Markers markers;
        markers.createNew("", 4096);
        {
            MarkerFinder finder(k, kmerTable, reads, markers, threadCount);
        }

        const uint64_t n = markers.size();
        *readCountOut = n / 2;
        uint64_t* toc = static_cast<uint64_t*>(std::malloc((n + 1) * sizeof(uint64_t)));
        toc[0] = 0;
        for(uint64_t i = 0; i < n; i++) toc[i+1] = toc[i] + markers.size(i);
        uint8_t* data = static_cast<uint8_t*>(std::malloc(std::max<uint64_t>(1, 7 * toc[n])));
        if(toc[n]) std::memcpy(data, markers.begin(), 7 * toc[n]);
        *tocOut = toc;
        *dataOut = data;
        markers.remove();
        kmerTable.remove();
        reads.remove();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}


// The same chain, also returning what the reference's MarkerFinder READ: the reads as stored
// (LongBaseSequences: two bit planes per 64 bases, src/LongBaseSequence.hpp:33-41), their base counts
// and the isMarker flag of every k-mer id.  For the parity tests of marker finding
// (MarkerFinder::MarkerFinder, src/MarkerFinder.cpp:16-127).
int ref_reads_and_markers_from_fasta(
    const char* fastaPath, uint64_t k, double probability, int seed,
    uint64_t minReadLength, uint64_t threadCount,
    uint64_t* readCountOut, uint64_t** readsTocOut, uint64_t** readsDataOut, uint64_t** baseCountsOut,
    uint8_t** isMarkerOut, uint64_t** tocOut, uint8_t** dataOut)
{
    try {
        CoutCapture capture;
        if(threadCount == 0) threadCount = std::thread::hardware_concurrency();
        Reads reads;
        reads.createNew(1, "", "", "", "", "", "", 4096);
        {
            ReadLoader loader(fastaPath, 1, minReadLength, false, threadCount, "", 4096, reads);
        }
        MemoryMapped::Vector<KmerInfo> kmerTable;
        kmerTable.createNew("", 4096);
        const uint64_t kmerCount = 1ULL << (2ULL * k);
        kmerTable.resize(kmerCount);
        for(uint64_t kmerId = 0; kmerId < kmerCount; kmerId++) {
            const Kmer kmer(kmerId, k);
            KmerInfo& info = kmerTable[kmerId];
            info.frequency = 0;
            info.reverseComplementedKmerId = KmerId(kmer.reverseComplement(k).id(k));
            info.isMarker = false;
            info.isRleKmer = true;
            info.hash = 0;
        }
        const double p = 1. - std::sqrt(1. - probability);
        std::mt19937 randomSource(seed);
        std::uniform_real_distribution<> uniformDistribution;
        for(uint64_t kmerId = 0; kmerId < kmerCount; kmerId++) {
            const double x = uniformDistribution(randomSource);
            if(x <= p) {
                kmerTable[kmerId].isMarker = true;
                kmerTable[kmerTable[kmerId].reverseComplementedKmerId].isMarker = true;
            }
        }
        Markers markers;
        markers.createNew("", 4096);
        {
            MarkerFinder finder(k, kmerTable, reads, markers, threadCount);
        }

        const uint64_t readCount = reads.readCount();
        *readCountOut = readCount;
        std::vector<uint64_t> readsToc(readCount + 1, 0), baseCounts(readCount), words;
        for(uint64_t r = 0; r < readCount; r++) {
            const LongBaseSequenceView v = reads.getRead(ReadId(r));
            baseCounts[r] = v.baseCount;
            const uint64_t n = LongBaseSequenceView::wordCount(v.baseCount);
            words.insert(words.end(), v.begin, v.begin + n);
            readsToc[r + 1] = words.size();
        }
        std::vector<uint8_t> isMarker(kmerCount);
        for(uint64_t i = 0; i < kmerCount; i++) isMarker[i] = kmerTable[i].isMarker ? 1 : 0;
        *readsTocOut = mallocCopy(readsToc); *readsDataOut = mallocCopy(words); *baseCountsOut = mallocCopy(baseCounts);
        *isMarkerOut = mallocCopy(isMarker);

        const uint64_t n = markers.size();
        std::vector<uint64_t> toc(n + 1, 0);
        for(uint64_t i = 0; i < n; i++) toc[i+1] = toc[i] + markers.size(i);
        std::vector<uint8_t> data(7 * toc[n]);
        if(toc[n]) std::memcpy(data.data(), markers.begin(), 7 * toc[n]);
        *tocOut = mallocCopy(toc); *dataOut = mallocCopy(data);
        markers.remove(); kmerTable.remove(); reads.remove();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}


// --------------------------------------------------------------------------
// Seam 1: the reference LowHash0, unmodified.
// --------------------------------------------------------------------------
int ref_lowhash0(
    uint64_t readCount,
    const uint64_t* markersToc,
    const void* markersData,
    const uint8_t* readFlags,
    const shasta_lowhash0_params* params,
    uint64_t threadCount,
    const char* workDirectory,      // the two CSV side files are written here
    uint64_t* readLowHashStatistics,
    shasta_lowhash0_result* result)
{
    try {
        std::memset(result, 0, sizeof(*result));
        ChdirGuard cd(workDirectory);
        Markers markers;
        fillMarkers(markers, readCount, markersToc, markersData);

        Reads reads;
        reads.createNew(1, "", "", "", "", "", "", 4096);
        reads.readFlags.resize(readCount);
        for(uint64_t i = 0; i < readCount; i++) {
            ReadFlags f;
            if(readFlags) *reinterpret_cast<uint8_t*>(&f) = readFlags[i];
            reads.readFlags[i] = f;
        }

        MemoryMapped::Vector<KmerInfo> kmerTable;      // never dereferenced by LowHash0
        kmerTable.createNew("", 4096);
        MemoryMapped::Vector<OrientedReadPair> candidates;
        candidates.createNew("", 4096);
        MemoryMapped::Vector< array<uint64_t, 3> > statistics;
        statistics.createNew("", 4096);

        std::string console;
        const auto t0 = std::chrono::steady_clock::now();
        {
            CoutCapture capture;
            LowHash0 lowHash0(
                params->m, params->hashFraction,
                params->minHashIterationCount, params->alignmentCandidatesPerRead,
                params->log2MinHashBucketCount,
                params->minBucketSize, params->maxBucketSize, params->minFrequency,
                threadCount, kmerTable, reads, markers, candidates, statistics, "", 4096);
            console = capture.str();
        }
        const auto t1 = std::chrono::steady_clock::now();
        result->seconds = std::chrono::duration<double>(t1 - t0).count();

        // Outputs.
        result->candidateCount = candidates.size();
        result->candidates = static_cast<shasta_oriented_read_pair*>(
            std::calloc(std::max<uint64_t>(1, candidates.size()), sizeof(shasta_oriented_read_pair)));
        for(uint64_t i = 0; i < candidates.size(); i++) {
            result->candidates[i].readIds[0] = candidates[i].readIds[0];
            result->candidates[i].readIds[1] = candidates[i].readIds[1];
            result->candidates[i].isSameStrand = candidates[i].isSameStrand ? 1 : 0;
        }
        for(uint64_t i = 0; i < readCount; i++) {
            for(int c = 0; c < 3; c++) readLowHashStatistics[3*i + c] = statistics[i][c];
        }

        // Console lines (src/LowHash0.cpp:97-98, :193-196).
        std::vector<uint64_t> high, total;
        {
            std::istringstream s(console);
            std::string line;
            while(std::getline(s, line)) {
                unsigned long long it, h, t, c;
                unsigned l2; unsigned long long bc;
                if(std::sscanf(line.c_str(),
                    "Alignment candidates after lowhash iteration %llu: high frequency %llu, total %llu, capacity %llu.",
                    &it, &h, &t, &c) == 4) {
                    high.push_back(h); total.push_back(t);
                } else if(std::sscanf(line.c_str(),
                    "LowHash0 algorithm will use 2^%u = %llu buckets.", &l2, &bc) == 2) {
                    result->log2BucketCount = l2;
                }
            }
        }
        result->iterationCount = uint32_t(high.size());
        result->highFrequency = mallocCopy(high);
        result->total = mallocCopy(total);

        // LowHashBucketHistogram.csv (src/LowHash0.cpp:586-595).
        std::vector<uint64_t> rows;
        {
            std::ifstream csv("LowHashBucketHistogram.csv");
            std::string line;
            std::getline(csv, line);
            while(std::getline(csv, line)) {
                unsigned long long a, b, c, d;
                if(std::sscanf(line.c_str(), "%llu,%llu,%llu,%llu", &a, &b, &c, &d) == 4) {
                    rows.push_back(a); rows.push_back(b); rows.push_back(c);
                }
            }
        }
        result->histogramRowCount = rows.size() / 3;
        result->histogram = mallocCopy(rows);

        candidates.remove();
        statistics.remove();
        kmerTable.remove();
        markers.remove();
        reads.remove();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

void ref_lowhash0_free(shasta_lowhash0_result* r)
{
    std::free(r->candidates); std::free(r->highFrequency); std::free(r->total); std::free(r->histogram);
    std::memset(r, 0, sizeof(*r));
}


// --------------------------------------------------------------------------
// Seam 2: per-candidate loop of src/AssemblerAlign.cpp:378-483 (method 4,
// one thread => output in candidate order) around the reference Align4::align.
// --------------------------------------------------------------------------
static void sortedMarkersOf(const span<const CompressedMarker>& um, vector< pair<KmerId, uint32_t> >& sm)
{
    // src/AssemblerAlign4.cpp:243-258
    const uint64_t n = um.size();
    sm.resize(n);
    for(uint64_t ordinal = 0; ordinal < n; ordinal++) {
        sm[ordinal] = make_pair(KmerId(um[ordinal].kmerId), uint32_t(ordinal));
    }
    sort(sm.begin(), sm.end(), OrderPairsByFirstOnly<KmerId, uint32_t>());
}

// Assembler::computeSortedMarkers (src/AssemblerAlign4.cpp:190-261), which the reference runs ONCE before its alignment
// threads start (src/AssemblerAlign.cpp:236-239): every oriented read's (kmerId, ordinal) pairs sorted by kmerId, in the
// reference's own container, filled by `threadCount` threads over batches of 10 000 oriented reads.  Kept here between
// ref_compute_sorted_markers and the ref_align4_batch_mt calls on the SAME markers (bench.py's CPU leg: the pass is timed
// on its own and the per-candidate loop reads spans of it, as computeAlignmentsThreadFunction does, :397-398); without it
// ref_align4_batch_mt sorts the two reads of every candidate itself (the small inputs of the tests).
namespace {
struct SortedMarkersCache {
    MemoryMapped::VectorOfVectors< pair<KmerId, uint32_t>, uint64_t > sorted;
    const void* markersData = nullptr;
    uint64_t markerCount = 0, readCount = 0;
    bool valid = false;
} sortedMarkersCache;
}  // namespace

int ref_compute_sorted_markers(uint64_t readCount, const uint64_t* markersToc, const void* markersData, uint64_t threadCount, double* seconds)
{
    try {
        const auto t0 = std::chrono::steady_clock::now();
        SortedMarkersCache& c = sortedMarkersCache;
        if(c.valid) { c.sorted.remove(); c.valid = false; }
        if(threadCount == 0) threadCount = std::thread::hardware_concurrency();
        const CompressedMarker* all = static_cast<const CompressedMarker*>(markersData);
        const uint64_t orientedReadCount = 2 * readCount;
        c.sorted.createNew("", 4096);
        c.sorted.beginPass1(orientedReadCount);
        for(uint64_t i = 0; i < orientedReadCount; i++) c.sorted.incrementCount(i, markersToc[i + 1] - markersToc[i]);
        c.sorted.beginPass2();
        c.sorted.endPass2(false);
        std::atomic<uint64_t> next(0);
        auto worker = [&]() {
            for(;;) {
                const uint64_t begin = next.fetch_add(10000);
                if(begin >= orientedReadCount) break;
                for(uint64_t i = begin; i < std::min(orientedReadCount, begin + 10000); i++) {
                    const span< pair<KmerId, uint32_t> > sm = c.sorted[i];
                    const CompressedMarker* m = all + markersToc[i];
                    for(uint32_t ordinal = 0; ordinal < sm.size(); ordinal++) { sm[ordinal].first = m[ordinal].kmerId; sm[ordinal].second = ordinal; }
                    sort(sm.begin(), sm.end(), OrderPairsByFirstOnly<KmerId, uint32_t>());
                }
            }
        };
        std::vector<std::thread> threads;
        for(uint64_t t = 1; t < threadCount; t++) threads.emplace_back(worker);
        worker();
        for(auto& t : threads) t.join();
        c.markersData = markersData; c.markerCount = markersToc[orientedReadCount]; c.readCount = readCount; c.valid = true;
        if(seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

void ref_drop_sorted_markers()
{
    if(sortedMarkersCache.valid) { sortedMarkersCache.sorted.remove(); sortedMarkersCache.valid = false; }
}

// Assembler::computeAlignmentTable (src/AssemblerAlign.cpp:509-571), the serial last step of computeAlignments, in the
// reference's own container: every alignment is listed under its two oriented reads and their reverse complements (two
// passes), then every oriented read's list is ordered by (the other oriented read, alignment index).
int ref_alignment_table(uint64_t readCount, uint64_t alignmentCount, const shasta_alignment_data* rows, uint64_t* tocOut, uint32_t* valuesOut, double* seconds)
{
    try {
        const auto t0 = std::chrono::steady_clock::now();
        const AlignmentData* alignmentData = reinterpret_cast<const AlignmentData*>(rows);
        MemoryMapped::VectorOfVectors<uint32_t, uint32_t> table;
        table.createNew("", 4096);
        table.beginPass1(ReadId(2 * readCount));
        auto fourReads = [&](const AlignmentData& ad, array<OrientedReadId, 4>& r) {
            r[0] = OrientedReadId(ad.readIds[0], 0);
            r[1] = OrientedReadId(ad.readIds[1], ad.isSameStrand ? 0 : 1);
            r[2] = r[0]; r[2].flipStrand();
            r[3] = r[1]; r[3].flipStrand();
        };
        array<OrientedReadId, 4> r;
        for(uint64_t i = 0; i < alignmentCount; i++) { fourReads(alignmentData[i], r); for(const OrientedReadId& o : r) table.incrementCount(o.getValue()); }
        table.beginPass2();
        for(uint32_t i = 0; i < alignmentCount; i++) { fourReads(alignmentData[i], r); for(const OrientedReadId& o : r) table.store(o.getValue(), i); }
        table.endPass2();
        vector< pair<OrientedReadId, uint32_t> > v;
        for(ReadId readId0 = 0; readId0 < readCount; readId0++) {
            for(Strand strand0 = 0; strand0 < 2; strand0++) {
                const OrientedReadId orientedReadId0(readId0, strand0);
                const span<uint32_t> section = table[orientedReadId0.getValue()];
                v.clear();
                for(const uint32_t alignmentIndex : section) v.push_back(make_pair(alignmentData[alignmentIndex].getOther(orientedReadId0), alignmentIndex));
                sort(v.begin(), v.end());
                for(size_t i = 0; i < v.size(); i++) section[i] = v[i].second;
            }
        }
        if(seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if(tocOut) { tocOut[0] = 0; for(uint64_t i = 0; i < 2 * readCount; i++) tocOut[i + 1] = tocOut[i] + table.size(i); }
        if(valuesOut && alignmentCount) std::memcpy(valuesOut, table.begin(), 4ULL * alignmentCount * sizeof(uint32_t));
        table.remove();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

static void copyInfo(const AlignmentInfo& info, shasta_alignment_info& out)
{
    std::memset(&out, 0, sizeof(out));
    for(int i = 0; i < 2; i++) {
        out.data[i].markerCount = info.data[i].markerCount;
        out.data[i].firstOrdinal = info.data[i].firstOrdinal;
        out.data[i].lastOrdinal = info.data[i].lastOrdinal;
    }
    out.markerCount = info.markerCount;
    out.minOrdinalOffset = info.minOrdinalOffset;
    out.maxOrdinalOffset = info.maxOrdinalOffset;
    out.averageOrdinalOffset = info.averageOrdinalOffset;
    out.maxSkip = info.maxSkip;
    out.maxDrift = info.maxDrift;
}

// The tie policy of the shimmed SeqAn call inside this library's Align4.cpp / AssemblerAlign3.cpp (shims/seqan/align.h ->
// oracle/banded_dp.hpp): 0 = the restated reading; the tie census runs the reference's control flow under the others.
void ref_set_tie_policy(int index) { oracle::activeTiePolicy() = oracle::tiePolicyByIndex(index); }

int ref_align4_batch_mt(
    uint64_t readCount,
    const uint64_t* markersToc,
    const void* markersData,
    uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates,
    const shasta_align4_options* o,
    int wantOrdinals,
    uint64_t threadCount,
    shasta_align4_result* result)
{
    try {
        std::memset(result, 0, sizeof(*result));
        const auto t0 = std::chrono::steady_clock::now();
        CoutCapture capture;
        const CompressedMarker* all = static_cast<const CompressedMarker*>(markersData);
        if(threadCount == 0) threadCount = std::thread::hardware_concurrency();

        Align4::Options options;
        options.deltaX = o->deltaX;
        options.deltaY = o->deltaY;
        options.minEntryCountPerCell = o->minEntryCountPerCell;
        options.maxDistanceFromBoundary = o->maxDistanceFromBoundary;
        options.minAlignedMarkerCount = o->minAlignedMarkerCount;
        options.minAlignedFraction = o->minAlignedFraction;
        options.maxSkip = o->maxSkip;
        options.maxDrift = o->maxDrift;
        options.maxTrim = o->maxTrim;
        options.maxBand = o->maxBand;
        options.matchScore = o->matchScore;
        options.mismatchScore = o->mismatchScore;
        options.gapScore = o->gapScore;

        struct PerCandidate { uint8_t status = SHASTA_ALIGN_EMPTY; AlignmentInfo info; Alignment alignment; };
        std::vector<PerCandidate> per(candidateCount);
        std::atomic<uint64_t> next(0);
        std::string firstError;
        std::mutex errorMutex;

        // One worker = the body of computeAlignmentsThreadFunction (src/AssemblerAlign.cpp:308-496)
        // for alignMethod 4: dynamic batches of 10 candidates (:243-249), its own 2 GiB arena (:353-355).
        auto worker = [&]() {
            try {
                MemoryMapped::ByteAllocator byteAllocator("", 4096, 2ULL * 1024 * 1024 * 1024);
                array<vector< pair<KmerId, uint32_t> >, 2> sorted;
                const SortedMarkersCache& cache = sortedMarkersCache;
                const bool precomputed = cache.valid && cache.markersData == markersData && cache.readCount == readCount && cache.markerCount == markersToc[2 * readCount];
                for(;;) {
                    const uint64_t begin = next.fetch_add(10);
                    if(begin >= candidateCount) break;
                    const uint64_t end = std::min(candidateCount, begin + 10);
                    for(uint64_t i = begin; i < end; i++) {
                        const shasta_oriented_read_pair& c = candidates[i];
                        SHASTA_ASSERT(c.readIds[0] < c.readIds[1]);
                        SHASTA_ASSERT(c.readIds[1] < readCount);
                        const OrientedReadId or0(c.readIds[0], 0);
                        const OrientedReadId or1(c.readIds[1], c.isSameStrand ? 0 : 1);
                        array<span<const CompressedMarker>, 2> m;
                        m[0] = span<const CompressedMarker>(all + markersToc[or0.getValue()], all + markersToc[or0.getValue() + 1]);
                        m[1] = span<const CompressedMarker>(all + markersToc[or1.getValue()], all + markersToc[or1.getValue() + 1]);
                        array<span< const pair<KmerId, uint32_t> >, 2> sm;
                        for(int j = 0; j < 2; j++) {
                            if(precomputed) {
                                const uint64_t o = (j == 0 ? or0 : or1).getValue();
                                sm[j] = span< const pair<KmerId, uint32_t> >(cache.sorted.begin(o), cache.sorted.end(o));
                                continue;
                            }
                            sortedMarkersOf(m[j], sorted[j]);
                            const pair<KmerId, uint32_t>* b = sorted[j].data();
                            sm[j] = span< const pair<KmerId, uint32_t> >(b, b + sorted[j].size());
                        }
                        PerCandidate& pc = per[i];
                        try {
                            Align4::align(m, sm, options, byteAllocator, pc.alignment, pc.info, false);
                            SHASTA_ASSERT(byteAllocator.isEmpty());
                        } catch(...) {
                            pc.status = SHASTA_ALIGN_SKIPPED;    // :419-435: skip this candidate
                            pc.alignment.clear();
                            continue;
                        }
                        const Alignment& alignment = pc.alignment;
                        const AlignmentInfo& alignmentInfo = pc.info;
                        pc.status = alignment.ordinals.empty() ? SHASTA_ALIGN_EMPTY : SHASTA_ALIGN_REJECTED;
                        // Filters, src/AssemblerAlign.cpp:439-472.
                        if(alignment.ordinals.size() < o->minAlignedMarkerCount) continue;
                        if(min(alignmentInfo.alignedFraction(0), alignmentInfo.alignedFraction(1)) < o->minAlignedFraction) continue;
                        uint32_t leftTrim, rightTrim;
                        tie(leftTrim, rightTrim) = alignmentInfo.computeTrim();
                        if(leftTrim > o->maxTrim || rightTrim > o->maxTrim) continue;
                        if(alignment.maxSkip() > o->maxSkip) continue;
                        if(alignment.maxDrift() > o->maxDrift) continue;
                        if(o->suppressContainments && alignmentInfo.isContaining(uint32_t(o->maxTrim))) continue;
                        pc.status = SHASTA_ALIGN_STORED;
                    }
                }
            } catch(std::exception& e) {
                std::lock_guard<std::mutex> lock(errorMutex);
                if(firstError.empty()) firstError = e.what();
            }
        };
        std::vector<std::thread> threads;
        for(uint64_t t = 1; t < threadCount; t++) threads.emplace_back(worker);
        worker();
        for(auto& t : threads) t.join();
        if(!firstError.empty()) throw std::runtime_error(firstError);

        // Gather in candidate order (the reference's order for threadCount == 1, :262-283).
        std::vector<shasta_alignment_data> alignmentData;
        std::vector<uint64_t> compressedToc(1, 0);
        std::vector<uint8_t> compressedData;
        std::vector<uint8_t> status(candidateCount, SHASTA_ALIGN_EMPTY);
        std::vector<uint64_t> ordinalsToc(1, 0);
        std::vector<uint32_t> ordinals;
        string compressedAlignment;
        for(uint64_t i = 0; i < candidateCount; i++) {
            const PerCandidate& pc = per[i];
            status[i] = pc.status;
            if(wantOrdinals) {
                for(const auto& p : pc.alignment.ordinals) { ordinals.push_back(p[0]); ordinals.push_back(p[1]); }
                ordinalsToc.push_back(ordinals.size() / 2);
            }
            if(pc.status != SHASTA_ALIGN_STORED) continue;
            shasta_alignment_data ad;
            std::memset(&ad, 0, sizeof(ad));
            ad.pair = candidates[i];
            ad.pair.isSameStrand = ad.pair.isSameStrand ? 1 : 0;
            ad.pair.pad[0] = ad.pair.pad[1] = ad.pair.pad[2] = 0;
            copyInfo(pc.info, ad.info);
            alignmentData.push_back(ad);
            shasta::compress(pc.alignment, compressedAlignment);
            compressedData.insert(compressedData.end(), compressedAlignment.begin(), compressedAlignment.end());
            compressedToc.push_back(compressedData.size());
        }

        result->alignmentCount = alignmentData.size();
        result->alignmentData = mallocCopy(alignmentData);
        result->compressedToc = mallocCopy(compressedToc);
        result->compressedData = mallocCopy(compressedData);
        result->status = mallocCopy(status);
        if(wantOrdinals) {
            result->ordinalsToc = mallocCopy(ordinalsToc);
            result->ordinals = mallocCopy(ordinals);
        }
        const auto t1 = std::chrono::steady_clock::now();
        result->seconds = std::chrono::duration<double>(t1 - t0).count();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

int ref_align4_batch(
    uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align4_options* o, int wantOrdinals, shasta_align4_result* result)
{
    return ref_align4_batch_mt(readCount, markersToc, markersData, candidateCount, candidates, o, wantOrdinals, 1, result);
}

// --------------------------------------------------------------------------
// Align method 3: the same per-candidate loop around the reference's own
// Assembler::alignOrientedReads3 (src/AssemblerAlign3.cpp, compiled in place by ref_align3.cpp).
// Non-reference code: the k-mer table fill (reverse complement and hash fields, restating
// src/AssemblerKmers.cpp:147-186) and the loop (src/AssemblerAlign.cpp:378-483).
// --------------------------------------------------------------------------
static void fillKmerTableForDownsampling(MemoryMapped::Vector<KmerInfo>& kmerTable, uint64_t k)
{
    kmerTable.createNew("", 4096);
    const uint64_t kmerCount = 1ULL << (2ULL * k);
    kmerTable.resize(kmerCount);
    for(uint64_t kmerId = 0; kmerId < kmerCount; kmerId++) {
        const Kmer kmer(kmerId, k);
        KmerInfo& info = kmerTable[kmerId];
        info.frequency = 0;
        info.reverseComplementedKmerId = KmerId(kmer.reverseComplement(k).id(k));
        info.isMarker = false;
        info.isRleKmer = false;
    }
    for(uint64_t kmerId = 0; kmerId < kmerCount; kmerId++) {
        const uint64_t n = kmerId + kmerTable[kmerId].reverseComplementedKmerId;    // :184
        kmerTable[kmerId].hash = MurmurHash2(&n, sizeof(n), 13477);                 // :185
    }
}

// KmerInfo::hash of every k-mer id (4^k values), for checking the restated hash.
int ref_kmer_hashes(uint64_t k, uint32_t* out)
{
    try {
        MemoryMapped::Vector<KmerInfo> kmerTable;
        fillKmerTableForDownsampling(kmerTable, k);
        for(uint64_t i = 0; i < kmerTable.size(); i++) out[i] = kmerTable[i].hash;
        kmerTable.remove();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

int ref_align3_batch_mt(
    uint64_t readCount,
    const uint64_t* markersToc,
    const void* markersData,
    uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates,
    const shasta_align3_options* o,
    int wantOrdinals,
    uint64_t threadCount,
    shasta_align4_result* result)
{
    try {
        std::memset(result, 0, sizeof(*result));
        const auto t0 = std::chrono::steady_clock::now();
        CoutCapture capture;
        if(threadCount == 0) threadCount = std::thread::hardware_concurrency();

        Assembler assembler;
        fillMarkers(assembler.markers, readCount, markersToc, markersData);
        fillKmerTableForDownsampling(assembler.kmerTable, o->k);

        struct PerCandidate { uint8_t status = SHASTA_ALIGN_EMPTY; AlignmentInfo info; Alignment alignment; };
        std::vector<PerCandidate> per(candidateCount);
        std::atomic<uint64_t> next(0);
        std::string firstError;
        std::mutex errorMutex;

        auto worker = [&]() {
            try {
                for(;;) {
                    const uint64_t begin = next.fetch_add(10);
                    if(begin >= candidateCount) break;
                    const uint64_t end = std::min(candidateCount, begin + 10);
                    for(uint64_t i = begin; i < end; i++) {
                        const shasta_oriented_read_pair& c = candidates[i];
                        SHASTA_ASSERT(c.readIds[0] < c.readIds[1]);
                        SHASTA_ASSERT(c.readIds[1] < readCount);
                        const OrientedReadId or0(c.readIds[0], 0);
                        const OrientedReadId or1(c.readIds[1], c.isSameStrand ? 0 : 1);
                        PerCandidate& pc = per[i];
                        try {
                            assembler.alignOrientedReads3(or0, or1,
                                int(o->matchScore), int(o->mismatchScore), int(o->gapScore),
                                o->downsamplingFactor, int(o->bandExtend), int(o->maxBand),
                                pc.alignment, pc.info);
                        } catch(...) {
                            pc.status = SHASTA_ALIGN_SKIPPED;    // :419-435: skip this candidate
                            pc.alignment.clear();
                            continue;
                        }
                        const Alignment& alignment = pc.alignment;
                        const AlignmentInfo& alignmentInfo = pc.info;
                        pc.status = alignment.ordinals.empty() ? SHASTA_ALIGN_EMPTY : SHASTA_ALIGN_REJECTED;
                        // Filters, src/AssemblerAlign.cpp:439-472.
                        if(alignment.ordinals.size() < o->minAlignedMarkerCount) continue;
                        if(min(alignmentInfo.alignedFraction(0), alignmentInfo.alignedFraction(1)) < o->minAlignedFraction) continue;
                        uint32_t leftTrim, rightTrim;
                        tie(leftTrim, rightTrim) = alignmentInfo.computeTrim();
                        if(leftTrim > o->maxTrim || rightTrim > o->maxTrim) continue;
                        if(alignment.maxSkip() > o->maxSkip) continue;
                        if(alignment.maxDrift() > o->maxDrift) continue;
                        if(o->suppressContainments && alignmentInfo.isContaining(uint32_t(o->maxTrim))) continue;
                        pc.status = SHASTA_ALIGN_STORED;
                    }
                }
            } catch(std::exception& e) {
                std::lock_guard<std::mutex> lock(errorMutex);
                if(firstError.empty()) firstError = e.what();
            }
        };
        std::vector<std::thread> threads;
        for(uint64_t t = 1; t < threadCount; t++) threads.emplace_back(worker);
        worker();
        for(auto& t : threads) t.join();
        assembler.markers.remove();
        assembler.kmerTable.remove();
        if(!firstError.empty()) throw std::runtime_error(firstError);

        std::vector<shasta_alignment_data> alignmentData;
        std::vector<uint64_t> compressedToc(1, 0);
        std::vector<uint8_t> compressedData;
        std::vector<uint8_t> status(candidateCount, SHASTA_ALIGN_EMPTY);
        std::vector<uint64_t> ordinalsToc(1, 0);
        std::vector<uint32_t> ordinals;
        string compressedAlignment;
        for(uint64_t i = 0; i < candidateCount; i++) {
            const PerCandidate& pc = per[i];
            status[i] = pc.status;
            if(wantOrdinals) {
                for(const auto& p : pc.alignment.ordinals) { ordinals.push_back(p[0]); ordinals.push_back(p[1]); }
                ordinalsToc.push_back(ordinals.size() / 2);
            }
            if(pc.status != SHASTA_ALIGN_STORED) continue;
            shasta_alignment_data ad;
            std::memset(&ad, 0, sizeof(ad));
            ad.pair = candidates[i];
            ad.pair.isSameStrand = ad.pair.isSameStrand ? 1 : 0;
            ad.pair.pad[0] = ad.pair.pad[1] = ad.pair.pad[2] = 0;
            copyInfo(pc.info, ad.info);
            alignmentData.push_back(ad);
            shasta::compress(pc.alignment, compressedAlignment);
            compressedData.insert(compressedData.end(), compressedAlignment.begin(), compressedAlignment.end());
            compressedToc.push_back(compressedData.size());
        }
        result->alignmentCount = alignmentData.size();
        result->alignmentData = mallocCopy(alignmentData);
        result->compressedToc = mallocCopy(compressedToc);
        result->compressedData = mallocCopy(compressedData);
        result->status = mallocCopy(status);
        if(wantOrdinals) {
            result->ordinalsToc = mallocCopy(ordinalsToc);
            result->ordinals = mallocCopy(ordinals);
        }
        result->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

void ref_align4_free(shasta_align4_result* r)
{
    std::free(r->alignmentData); std::free(r->compressedToc); std::free(r->compressedData);
    std::free(r->status); std::free(r->ordinalsToc); std::free(r->ordinals);
    std::memset(r, 0, sizeof(*r));
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Data/ directory fixtures written and checked by the reference's own containers
// (MemoryMapped::Vector / VectorOfVectors), for the host layer's file-format tests.
// ---------------------------------------------------------------------------
namespace {
template<size_t N> struct Blob { char bytes[N]; };
template<size_t N> int openBlobVector(const std::string& path, uint64_t* objectCount, uint64_t* fileSize, void* out, uint64_t outCapacity)
{
    MemoryMapped::Vector< Blob<N> > v;
    v.accessExistingReadOnly(path);
    *objectCount = v.size();
    *fileSize = 4096 + N * v.capacity();
    if(out) {
        if(N * v.size() > outCapacity) throw std::runtime_error("output capacity too small");
        if(v.size()) std::memcpy(out, v.begin(), N * v.size());
    }
    return 0;
}
}  // namespace

extern "C" {

int ref_write_data_dir(const char* dir, uint64_t readCount, const uint64_t* toc, const void* data7, const uint8_t* flags)
{
    try {
        const std::string d(dir);
        Markers markers;
        markers.createNew(d + "/Markers", 4096);
        for(uint64_t i = 0; i < 2 * readCount; i++) {
            const CompressedMarker* b = reinterpret_cast<const CompressedMarker*>(static_cast<const char*>(data7) + 7 * toc[i]);
            markers.appendVector(b, b + (toc[i + 1] - toc[i]));
        }
        markers.unreserve();
        MemoryMapped::Vector<ReadFlags> readFlags;
        readFlags.createNew(d + "/ReadFlags", 4096);
        readFlags.resize(readCount);
        for(uint64_t i = 0; i < readCount; i++) {
            ReadFlags f;
            if(flags) *reinterpret_cast<uint8_t*>(&f) = flags[i];
            readFlags[i] = f;
        }
        readFlags.unreserve();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

// Opens <path> with MemoryMapped::Vector<T>::accessExistingReadOnly for a T of objectSize bytes
// (the reference's magic number / file size / object size checks) and optionally copies the data.
int ref_open_vector(const char* path, uint64_t objectSize, uint64_t* objectCount, uint64_t* fileSize, void* out, uint64_t outCapacity)
{
    try {
        switch(objectSize) {
            case 1: return openBlobVector<1>(path, objectCount, fileSize, out, outCapacity);
            case 4: return openBlobVector<4>(path, objectCount, fileSize, out, outCapacity);
            case 7: return openBlobVector<7>(path, objectCount, fileSize, out, outCapacity);
            case 8: return openBlobVector<8>(path, objectCount, fileSize, out, outCapacity);
            case 12: return openBlobVector<12>(path, objectCount, fileSize, out, outCapacity);
            case 16: return openBlobVector<16>(path, objectCount, fileSize, out, outCapacity);
            case 24: return openBlobVector<24>(path, objectCount, fileSize, out, outCapacity);
            case 64: return openBlobVector<64>(path, objectCount, fileSize, out, outCapacity);
            default: throw std::runtime_error("unsupported object size");
        }
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

// The reference's LowHash0 over an existing Data/ directory: inputs opened from Markers /
// ReadFlags, outputs created as AlignmentCandidates / ReadLowHashStatistics exactly as
// Assembler::findAlignmentCandidatesLowHash0 does (src/AssemblerLowHash.cpp:25-54).
int ref_lowhash0_files(const char* dir, const shasta_lowhash0_params* params, uint64_t threadCount, const char* workDirectory)
{
    try {
        const std::string d(dir);
        ChdirGuard cd(workDirectory);
        Markers markers;
        markers.accessExistingReadOnly(d + "/Markers");
        MemoryMapped::Vector<ReadFlags> flagsFile;
        flagsFile.accessExistingReadOnly(d + "/ReadFlags");
        const uint64_t readCount = markers.size() / 2;
        Reads reads;
        reads.createNew(1, "", "", "", "", "", "", 4096);
        reads.readFlags.resize(readCount);
        for(uint64_t i = 0; i < readCount; i++) reads.readFlags[i] = flagsFile[i];
        MemoryMapped::Vector<KmerInfo> kmerTable;
        kmerTable.createNew("", 4096);
        MemoryMapped::Vector<OrientedReadPair> candidates;
        candidates.createNew(d + "/AlignmentCandidates", 4096);
        MemoryMapped::Vector< array<uint64_t, 3> > statistics;
        statistics.createNew(d + "/ReadLowHashStatistics", 4096);
        {
            CoutCapture capture;
            LowHash0 lowHash0(
                params->m, params->hashFraction,
                params->minHashIterationCount, params->alignmentCandidatesPerRead,
                params->log2MinHashBucketCount,
                params->minBucketSize, params->maxBucketSize, params->minFrequency,
                threadCount, kmerTable, reads, markers, candidates, statistics, d + "/", 4096);
            std::ofstream console("LowHash0.console");
            console << capture.str();
        }
        candidates.unreserve();
        statistics.unreserve();
        kmerTable.remove();
        reads.remove();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

// AlignmentData + CompressedAlignments written the way Assembler::computeAlignments stores them
// (src/AssemblerAlign.cpp:262-287).
int ref_store_alignments(const char* dir, uint64_t alignmentCount, const shasta_alignment_data* rows,
    const uint64_t* compressedToc, const uint8_t* compressedData)
{
    try {
        const std::string d(dir);
        MemoryMapped::Vector<AlignmentData> alignmentData;
        alignmentData.createNew(d + "/AlignmentData", 4096);
        MemoryMapped::VectorOfVectors<char, uint64_t> compressedAlignments;
        compressedAlignments.createNew(d + "/CompressedAlignments", 4096);
        for(uint64_t i = 0; i < alignmentCount; i++) {
            AlignmentData ad;
            std::memcpy(&ad, rows + i, sizeof(ad));
            alignmentData.push_back(ad);
            const char* b = reinterpret_cast<const char*>(compressedData) + compressedToc[i];
            compressedAlignments.appendVector(b, b + (compressedToc[i + 1] - compressedToc[i]));
        }
        alignmentData.unreserve();
        compressedAlignments.unreserve();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

}  // extern "C"
