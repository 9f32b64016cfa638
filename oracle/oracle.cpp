// TEST INFRASTRUCTURE ONLY -- builds into oracle/_build/liboracle.so.
// C ABI over the CPU restatement in restated.hpp / banded_dp.hpp, with the same
// struct layouts as include/shasta_mi355x.h so that tests can diff the product
// against it field by field.  Used by tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg ONLY.
#include "restated.hpp"
#include "method0.hpp"
#include "../include/shasta_mi355x.h"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <string>
#include <thread>

using namespace oracle;

static thread_local std::string lastError;
static std::atomic<uint64_t> threadCountSetting(1);

template<class T> static T* mallocCopy(const std::vector<T>& v)
{
    T* p = static_cast<T*>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
    if(!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

static void copyInfo(const Info& info, shasta_alignment_info& out)
{
    std::memset(&out, 0, sizeof(out));
    for(int i = 0; i < 2; i++) {
        out.data[i].markerCount = info.markerCounts[i];
        out.data[i].firstOrdinal = info.firstOrdinal[i];
        out.data[i].lastOrdinal = info.lastOrdinal[i];
    }
    out.markerCount = info.markerCount;
    out.minOrdinalOffset = info.minOrdinalOffset;
    out.maxOrdinalOffset = info.maxOrdinalOffset;
    out.averageOrdinalOffset = info.averageOrdinalOffset;
    out.maxSkip = info.maxSkip;
    out.maxDrift = info.maxDrift;
}

extern "C" {

const char* oracle_last_error() { return lastError.c_str(); }
void oracle_free(void* p) { std::free(p); }
// The DP tie policy of this process's oracle (banded_dp.hpp, tiePolicyByIndex; 0 = the restated reading).  Set between runs, never during one.
void oracle_set_tie_policy(int index) { activeTiePolicy() = tiePolicyByIndex(index); }
void oracle_set_threads(uint64_t n) { threadCountSetting = n ? n : std::thread::hardware_concurrency(); }

uint64_t oracle_murmur64a(const void* key, int len, uint64_t seed) { return murmurHash64A(key, len, seed); }

int oracle_hash_windows(const uint32_t* kmerIds, uint64_t n, uint64_t m, uint64_t iteration, uint64_t* out)
{
    for(uint64_t j = 0; j + m <= n; j++) out[j] = murmurHash64A(kmerIds + j, int(4 * m), 37 * iteration);
    return 0;
}

int oracle_lowhash0(
    uint64_t readCount, const uint64_t* markersToc, const void* markersData, const uint8_t* readFlags,
    const shasta_lowhash0_params* params, uint64_t /*threadCount*/,
    uint64_t* readLowHashStatistics, shasta_lowhash0_result* result)
{
    try {
        std::memset(result, 0, sizeof(*result));
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<uint32_t> kmerIds;
        extractKmerIds(static_cast<const uint8_t*>(markersData), markersToc[2 * readCount], kmerIds);
        LowHash0Params p;
        p.m = params->m; p.hashFraction = params->hashFraction;
        p.minHashIterationCount = params->minHashIterationCount;
        p.alignmentCandidatesPerRead = params->alignmentCandidatesPerRead;
        p.log2MinHashBucketCount = params->log2MinHashBucketCount;
        p.minBucketSize = params->minBucketSize; p.maxBucketSize = params->maxBucketSize;
        p.minFrequency = params->minFrequency;
        LowHash0Output out;
        lowHash0(readCount, markersToc, kmerIds.data(), readFlags, p, out);

        result->candidateCount = out.candidates.size();
        result->candidates = static_cast<shasta_oriented_read_pair*>(
            std::calloc(std::max<size_t>(1, out.candidates.size()), sizeof(shasta_oriented_read_pair)));
        for(size_t i = 0; i < out.candidates.size(); i++) {
            result->candidates[i].readIds[0] = out.candidates[i][0];
            result->candidates[i].readIds[1] = out.candidates[i][1];
            result->candidates[i].isSameStrand = uint8_t(out.candidates[i][2]);
        }
        for(uint64_t i = 0; i < readCount; i++) for(int c = 0; c < 3; c++) {
            readLowHashStatistics[3 * i + c] = out.statistics[i][c];
        }
        result->log2BucketCount = out.log2BucketCount;
        result->iterationCount = uint32_t(out.highFrequency.size());
        result->highFrequency = mallocCopy(out.highFrequency);
        result->total = mallocCopy(out.total);
        std::vector<uint64_t> rows;
        for(const auto& r : out.histogram) { rows.push_back(r[0]); rows.push_back(r[1]); rows.push_back(r[2]); }
        result->histogramRowCount = out.histogram.size();
        result->histogram = mallocCopy(rows);
        result->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

void oracle_lowhash0_free(shasta_lowhash0_result* r)
{
    std::free(r->candidates); std::free(r->highFrequency); std::free(r->total); std::free(r->histogram);
    std::memset(r, 0, sizeof(*r));
}

int oracle_align4_batch(
    uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align4_options* o, int wantOrdinals, shasta_align4_result* result)
{
    try {
        std::memset(result, 0, sizeof(*result));
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<uint32_t> kmerIds;
        extractKmerIds(static_cast<const uint8_t*>(markersData), markersToc[2 * readCount], kmerIds);
        Align4Options opt;
        opt.deltaX = o->deltaX; opt.deltaY = o->deltaY;
        opt.minEntryCountPerCell = o->minEntryCountPerCell;
        opt.maxDistanceFromBoundary = o->maxDistanceFromBoundary;
        opt.minAlignedMarkerCount = o->minAlignedMarkerCount;
        opt.minAlignedFraction = o->minAlignedFraction;
        opt.maxSkip = o->maxSkip; opt.maxDrift = o->maxDrift; opt.maxTrim = o->maxTrim; opt.maxBand = o->maxBand;

        struct PerCandidate { Ordinals ord; Info info; uint8_t status; uint64_t dpCells; };
        std::vector<PerCandidate> per(candidateCount);
        const uint64_t threadCount = std::max<uint64_t>(1, threadCountSetting.load());
        std::atomic<uint64_t> next(0);
        auto work = [&]() {
            Align4Trace trace;
            for(;;) {
                const uint64_t begin = next.fetch_add(10);        // batches of 10, src/AssemblerAlign.cpp:243
                if(begin >= candidateCount) break;
                const uint64_t end = std::min(candidateCount, begin + 10);
                for(uint64_t i = begin; i < end; i++) {
                    const auto& c = candidates[i];
                    const uint64_t or0 = 2ULL * c.readIds[0];                       // strand 0, :382
                    const uint64_t or1 = 2ULL * c.readIds[1] + (c.isSameStrand ? 0 : 1);
                    const uint32_t nx = uint32_t(markersToc[or0 + 1] - markersToc[or0]);
                    const uint32_t ny = uint32_t(markersToc[or1 + 1] - markersToc[or1]);
                    PerCandidate& pc = per[i];
                    align4(kmerIds.data() + markersToc[or0], nx, kmerIds.data() + markersToc[or1], ny,
                        opt, pc.ord, pc.info, &trace);
                    pc.dpCells = trace.dpCells;
                    if(pc.ord.empty()) pc.status = SHASTA_ALIGN_EMPTY;
                    else pc.status = passesOuterFilters(pc.ord, pc.info, opt, o->suppressContainments != 0) ?
                        SHASTA_ALIGN_STORED : SHASTA_ALIGN_REJECTED;
                    if(trace.tie) pc.status |= SHASTA_ALIGN_TIE_FLAG;
                }
            }
        };
        std::vector<std::thread> threads;
        for(uint64_t t = 1; t < threadCount; t++) threads.emplace_back(work);
        work();
        for(auto& t : threads) t.join();

        std::vector<shasta_alignment_data> alignmentData;
        std::vector<uint64_t> compressedToc(1, 0), ordinalsToc(1, 0);
        std::vector<uint8_t> compressedData, bytes, status(candidateCount);
        std::vector<uint32_t> ordinals;
        for(uint64_t i = 0; i < candidateCount; i++) {
            const PerCandidate& pc = per[i];
            status[i] = pc.status;
            result->dpCellCount += pc.dpCells;
            const auto& c = candidates[i];
            const uint64_t or0 = 2ULL * c.readIds[0], or1 = 2ULL * c.readIds[1] + (c.isSameStrand ? 0 : 1);
            result->kmerIdBytes += 4 * (markersToc[or0 + 1] - markersToc[or0] + markersToc[or1 + 1] - markersToc[or1]);
            if(wantOrdinals) {
                for(const auto& p : pc.ord) { ordinals.push_back(p.first); ordinals.push_back(p.second); }
                ordinalsToc.push_back(ordinals.size() / 2);
            }
            if((pc.status & 0x7f) != SHASTA_ALIGN_STORED) continue;
            shasta_alignment_data ad;
            std::memset(&ad, 0, sizeof(ad));
            ad.pair.readIds[0] = c.readIds[0]; ad.pair.readIds[1] = c.readIds[1];
            ad.pair.isSameStrand = c.isSameStrand ? 1 : 0;
            copyInfo(pc.info, ad.info);
            alignmentData.push_back(ad);
            compress(pc.ord, bytes);
            compressedData.insert(compressedData.end(), bytes.begin(), bytes.end());
            compressedToc.push_back(compressedData.size());
        }
        result->alignmentCount = alignmentData.size();
        result->alignmentData = mallocCopy(alignmentData);
        result->compressedToc = mallocCopy(compressedToc);
        result->compressedData = mallocCopy(compressedData);
        result->status = mallocCopy(status);
        if(wantOrdinals) { result->ordinalsToc = mallocCopy(ordinalsToc); result->ordinals = mallocCopy(ordinals); }
        result->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

// Align method 3 (restated.hpp: align3).  Same result layout as the Align4 batch; status EMPTY
// = empty alignment, SKIPPED = the reference's exception lane.
int oracle_align3_batch(
    uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align3_options* o, int wantOrdinals, shasta_align4_result* result)
{
    try {
        std::memset(result, 0, sizeof(*result));
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<uint32_t> kmerIds;
        extractKmerIds(static_cast<const uint8_t*>(markersData), markersToc[2 * readCount], kmerIds);
        Align3Options opt;
        opt.matchScore = int32_t(o->matchScore); opt.mismatchScore = int32_t(o->mismatchScore); opt.gapScore = int32_t(o->gapScore);
        opt.downsamplingFactor = o->downsamplingFactor;
        opt.bandExtend = int32_t(o->bandExtend); opt.maxBand = int32_t(o->maxBand); opt.k = o->k;
        Align4Options filters;
        filters.minAlignedMarkerCount = o->minAlignedMarkerCount;
        filters.minAlignedFraction = o->minAlignedFraction;
        filters.maxSkip = o->maxSkip; filters.maxDrift = o->maxDrift; filters.maxTrim = o->maxTrim;

        struct PerCandidate { Ordinals ord; Info info; uint8_t status; uint64_t dpCells; };
        std::vector<PerCandidate> per(candidateCount);
        const uint64_t threadCount = std::max<uint64_t>(1, threadCountSetting.load());
        std::atomic<uint64_t> next(0);
        auto work = [&]() {
            Align3Trace trace;
            for(;;) {
                const uint64_t begin = next.fetch_add(10);
                if(begin >= candidateCount) break;
                const uint64_t end = std::min(candidateCount, begin + 10);
                for(uint64_t i = begin; i < end; i++) {
                    const auto& c = candidates[i];
                    const uint64_t or0 = 2ULL * c.readIds[0];
                    const uint64_t or1 = 2ULL * c.readIds[1] + (c.isSameStrand ? 0 : 1);
                    const uint32_t nx = uint32_t(markersToc[or0 + 1] - markersToc[or0]);
                    const uint32_t ny = uint32_t(markersToc[or1 + 1] - markersToc[or1]);
                    PerCandidate& pc = per[i];
                    const bool ok = align3(kmerIds.data() + markersToc[or0], nx, kmerIds.data() + markersToc[or1], ny,
                        opt, pc.ord, pc.info, &trace);
                    pc.dpCells = trace.dpCells;
                    if(!ok) { pc.status = SHASTA_ALIGN_SKIPPED; pc.ord.clear(); }
                    else if(pc.ord.empty()) pc.status = SHASTA_ALIGN_EMPTY;
                    else pc.status = passesOuterFilters(pc.ord, pc.info, filters, o->suppressContainments != 0) ?
                        SHASTA_ALIGN_STORED : SHASTA_ALIGN_REJECTED;
                }
            }
        };
        std::vector<std::thread> threads;
        for(uint64_t t = 1; t < threadCount; t++) threads.emplace_back(work);
        work();
        for(auto& t : threads) t.join();

        std::vector<shasta_alignment_data> alignmentData;
        std::vector<uint64_t> compressedToc(1, 0), ordinalsToc(1, 0);
        std::vector<uint8_t> compressedData, bytes, status(candidateCount);
        std::vector<uint32_t> ordinals;
        for(uint64_t i = 0; i < candidateCount; i++) {
            const PerCandidate& pc = per[i];
            status[i] = pc.status;
            result->dpCellCount += pc.dpCells;
            const auto& c = candidates[i];
            const uint64_t or0 = 2ULL * c.readIds[0], or1 = 2ULL * c.readIds[1] + (c.isSameStrand ? 0 : 1);
            result->kmerIdBytes += 4 * (markersToc[or0 + 1] - markersToc[or0] + markersToc[or1 + 1] - markersToc[or1]);
            if(wantOrdinals) {
                for(const auto& p : pc.ord) { ordinals.push_back(p.first); ordinals.push_back(p.second); }
                ordinalsToc.push_back(ordinals.size() / 2);
            }
            if(pc.status != SHASTA_ALIGN_STORED) continue;
            shasta_alignment_data ad;
            std::memset(&ad, 0, sizeof(ad));
            ad.pair.readIds[0] = c.readIds[0]; ad.pair.readIds[1] = c.readIds[1];
            ad.pair.isSameStrand = c.isSameStrand ? 1 : 0;
            copyInfo(pc.info, ad.info);
            alignmentData.push_back(ad);
            compress(pc.ord, bytes);
            compressedData.insert(compressedData.end(), bytes.begin(), bytes.end());
            compressedToc.push_back(compressedData.size());
        }
        result->alignmentCount = alignmentData.size();
        result->alignmentData = mallocCopy(alignmentData);
        result->compressedToc = mallocCopy(compressedToc);
        result->compressedData = mallocCopy(compressedData);
        result->status = mallocCopy(status);
        if(wantOrdinals) { result->ordinalsToc = mallocCopy(ordinalsToc); result->ordinals = mallocCopy(ordinals); }
        result->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

// Stage products of method 3 for one pair of kmer-id sequences: out[8] = {downsampled count 0, 1,
// step 1 aligned (0/1), offsetMin, offsetMax, bandMin, bandMax, band too wide (0/1)}.
int oracle_align3_stages(const uint32_t* k0, uint32_t nx, const uint32_t* k1, uint32_t ny,
    const shasta_align3_options* o, int64_t* out)
{
    try {
        Align3Options opt;
        opt.matchScore = int32_t(o->matchScore); opt.mismatchScore = int32_t(o->mismatchScore); opt.gapScore = int32_t(o->gapScore);
        opt.downsamplingFactor = o->downsamplingFactor;
        opt.bandExtend = int32_t(o->bandExtend); opt.maxBand = int32_t(o->maxBand); opt.k = o->k;
        Ordinals ord; Info info; Align3Trace t;
        align3(k0, nx, k1, ny, opt, ord, info, &t);
        out[0] = t.downsampledCount[0]; out[1] = t.downsampledCount[1]; out[2] = t.downsampledAligned ? 1 : 0;
        out[3] = t.offsetMin; out[4] = t.offsetMax; out[5] = t.bandMin; out[6] = t.bandMax; out[7] = t.bandTooWide ? 1 : 0;
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

// KmerInfo::hash of kmerIds[i] for marker length k.
int oracle_kmer_hashes(const uint32_t* kmerIds, uint64_t n, uint64_t k, uint32_t* out)
{
    for(uint64_t i = 0; i < n; i++) out[i] = kmerDownsamplingHash(kmerIds[i], k);
    return 0;
}

// Marker finding for readCount reads (restated.hpp: findMarkersOfRead).  toc gets 2*readCount+1
// entries; *data is malloc'ed (7 bytes per marker; free with oracle_free).
int oracle_find_markers(uint64_t readCount, const uint64_t* readsToc, const uint64_t* readsData, const uint64_t* baseCounts,
    uint64_t k, const uint8_t* isMarker, uint64_t* toc, uint8_t** data)
{
    try {
        std::vector<uint8_t> all, s0, s1;
        toc[0] = 0;
        for(uint64_t r = 0; r < readCount; r++) {
            findMarkersOfRead(readsData + readsToc[r], baseCounts[r], k, isMarker, s0, s1);
            all.insert(all.end(), s0.begin(), s0.end());
            toc[2 * r + 1] = all.size() / 7;
            all.insert(all.end(), s1.begin(), s1.end());
            toc[2 * r + 2] = all.size() / 7;
        }
        *data = mallocCopy(all);
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

void oracle_align4_free(shasta_align4_result* r)
{
    std::free(r->alignmentData); std::free(r->compressedToc); std::free(r->compressedData);
    std::free(r->status); std::free(r->ordinalsToc); std::free(r->ordinals);
    std::memset(r, 0, sizeof(*r));
}

// Stage seam: cells and bands of one pair.  cells rows: iX, iY, flags
// (bit0 nearLeftOrTop, bit1 nearRightOrBottom, bit2 forward, bit3 backward).
int oracle_align4_cells(
    const uint32_t* k0, uint32_t nx, const uint32_t* k1, uint32_t ny,
    const shasta_align4_options* o,
    uint32_t** cellsOut, uint64_t* cellCount, int32_t** bandsOut, uint64_t* bandCount)
{
    try {
        Align4Options opt;
        opt.deltaX = o->deltaX; opt.deltaY = o->deltaY;
        opt.minEntryCountPerCell = o->minEntryCountPerCell;
        opt.maxDistanceFromBoundary = o->maxDistanceFromBoundary;
        opt.minAlignedMarkerCount = o->minAlignedMarkerCount;
        opt.minAlignedFraction = o->minAlignedFraction;
        opt.maxSkip = o->maxSkip; opt.maxDrift = o->maxDrift; opt.maxTrim = o->maxTrim; opt.maxBand = o->maxBand;
        Ordinals ord; Info info; Align4Trace trace;
        align4(k0, nx, k1, ny, opt, ord, info, &trace);
        std::vector<uint32_t> cells;
        for(const Cell& c : trace.cells) {
            cells.push_back(c.iX); cells.push_back(c.iY);
            cells.push_back(uint32_t(c.nearLeftOrTop) | uint32_t(c.nearRightOrBottom) << 1 |
                uint32_t(c.forward) << 2 | uint32_t(c.backward) << 3);
        }
        std::vector<int32_t> bands;
        for(const auto& b : trace.bands) { bands.push_back(b.first); bands.push_back(b.second); }
        *cellCount = trace.cells.size(); *cellsOut = mallocCopy(cells);
        *bandCount = trace.bands.size(); *bandsOut = mallocCopy(bands);
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

int oracle_banded_dp(
    const uint32_t* k0, uint32_t nx, const uint32_t* k1, uint32_t ny,
    int32_t bandMin, int32_t bandMax,
    uint32_t* ordinals, uint64_t capacity, uint64_t* count, int32_t* score)
{
    try {
        BandedDpResult dp;
        bandedOverlapAlignment(k0, nx, k1, ny, 6, -1, -1, bandMin, bandMax, dp);
        Ordinals ord;
        diagonalMatches(k0, k1, dp, ord);
        if(ord.size() > capacity) throw std::runtime_error("oracle_banded_dp: capacity");
        for(size_t i = 0; i < ord.size(); i++) { ordinals[2*i] = ord[i].first; ordinals[2*i+1] = ord[i].second; }
        *count = ord.size();
        *score = dp.ok ? dp.score : std::numeric_limits<int32_t>::min();
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

// The same task from its matches (oracle/sparse_chain.hpp).  out[0] certified, [1] score, [2] hits, [3] scan steps, [4] reason.
int oracle_sparse_dp(
    const uint32_t* k0, uint32_t nx, const uint32_t* k1, uint32_t ny, int32_t bandMin, int32_t bandMax, uint32_t scanBudgetPerHit,
    uint32_t* ordinals, uint64_t capacity, uint64_t* count, int64_t* out)
{
    try {
        SparseChainResult r;
        sparseChainAlignment(k0, nx, k1, ny, bandMin, bandMax, r, scanBudgetPerHit);
        if(r.ordinals.size() > capacity) throw std::runtime_error("oracle_sparse_dp: capacity");
        for(size_t i = 0; i < r.ordinals.size(); i++) { ordinals[2*i] = r.ordinals[i].first; ordinals[2*i+1] = r.ordinals[i].second; }
        *count = r.ordinals.size();
        out[0] = r.certified ? 1 : 0; out[1] = r.score; out[2] = int64_t(r.hits); out[3] = int64_t(r.scanSteps); out[4] = r.reason;
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

// The same task with the dense DP confined to the stretches where the optimal chains differ (oracle/anchored_chain.hpp), under the
// policy in force.  out[0] score, [1] hits, [2] anchors, [3] windows, [4] dense cells solved, [5] whole task dense, [6] cells of the largest rectangle.
int oracle_anchored_dp(
    const uint32_t* k0, uint32_t nx, const uint32_t* k1, uint32_t ny, int32_t bandMin, int32_t bandMax,
    uint32_t* ordinals, uint64_t capacity, uint64_t* count, int64_t* out)
{
    try {
        AnchoredResult r;
        anchoredAlignment(k0, nx, k1, ny, bandMin, bandMax, r);
        if(r.ordinals.size() > capacity) throw std::runtime_error("oracle_anchored_dp: capacity");
        for(size_t i = 0; i < r.ordinals.size(); i++) { ordinals[2*i] = r.ordinals[i].first; ordinals[2*i+1] = r.ordinals[i].second; }
        *count = r.ordinals.size();
        out[0] = r.score; out[1] = int64_t(r.hits); out[2] = int64_t(r.anchors); out[3] = int64_t(r.windows); out[4] = int64_t(r.denseCells); out[5] = r.wholeTaskDense ? 1 : 0; out[6] = int64_t(r.largestWindow);
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

// What the sparse path would have done on every DP task the restated Align4 ran since the last reset (oracle_sparse_census(1)
// switches the bookkeeping on, (0) off): [0] tasks, [1] certified, [2] certified and DIFFERENT from the dense DP under the policy
// in force (must stay 0), [3] dense cells nx x width, [4] hits, [5] scan steps, [6..9] tasks by reason 0..3,
// [10] dense cells of the certified tasks, [11] aligned pairs, [12] tasks by reason 4.
void oracle_sparse_census(int on) { sparseCensus().on = on != 0; }
void oracle_sparse_census_anchored(int on) { sparseCensus().anchored = on != 0; }
void oracle_sparse_census_options(uint32_t scanBudget, int oneHitPerMarker, int runningMaxBound)
{
    sparseCensus().scanBudget = scanBudget; sparseCensus().oneHitPerMarker = oneHitPerMarker != 0; sparseCensus().runningMaxBound = runningMaxBound != 0;
}
void oracle_sparse_census_reset() { for(auto& c : sparseCensus().counters) c.store(0); }
void oracle_sparse_census_read(uint64_t* out) { for(int i = 0; i < SparseCensus::N; i++) out[i] = sparseCensus().counters[i].load(); }

int oracle_compress(const uint32_t* ordinals, uint64_t n, uint8_t** bytes, uint64_t* byteCount)
{
    Ordinals ord(n);
    for(uint64_t i = 0; i < n; i++) ord[i] = std::make_pair(ordinals[2*i], ordinals[2*i+1]);
    std::vector<uint8_t> s;
    compress(ord, s);
    *byteCount = s.size();
    *bytes = mallocCopy(s);
    return 0;
}

int oracle_decompress(const uint8_t* bytes, uint64_t byteCount, uint32_t** ordinals, uint64_t* n)
{
    Ordinals ord;
    decompress(bytes, byteCount, ord);
    std::vector<uint32_t> flat;
    for(const auto& p : ord) { flat.push_back(p.first); flat.push_back(p.second); }
    *n = ord.size();
    *ordinals = mallocCopy(flat);
    return 0;
}

int oracle_alignment_info(const uint32_t* ordinals, uint64_t n, uint32_t nx, uint32_t ny, shasta_alignment_info* out)
{
    Ordinals ord(n);
    for(uint64_t i = 0; i < n; i++) ord[i] = std::make_pair(ordinals[2*i], ordinals[2*i+1]);
    Info info;
    createInfo(ord, nx, ny, info);
    copyInfo(info, *out);
    return 0;
}

// Palindromic-read flagging: method-0 self-alignment of every read + the flag rule.
// flags / alignedCount / nearDiagonalCount / digests: readCount entries each (the last three optional).
int oracle_flag_palindromic_reads(
    uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    uint32_t maxSkip, uint32_t maxDrift, uint32_t maxMarkerFrequency,
    double alignedFractionThreshold, double nearDiagonalFractionThreshold, uint32_t deltaThreshold,
    uint8_t* flags, uint32_t* alignedCount, uint32_t* nearDiagonalCount, uint64_t* digests)
{
    try {
        const uint8_t* all = static_cast<const uint8_t*>(markersData);
        std::array<std::vector<uint32_t>, 2> kmerIds;
        std::vector<std::array<uint32_t, 2>> alignment;
        for(uint64_t r = 0; r < readCount; r++) {
            for(uint64_t s = 0; s < 2; s++) {
                const uint64_t begin = markersToc[2 * r + s], end = markersToc[2 * r + s + 1];
                extractKmerIds(all + 7 * begin, end - begin, kmerIds[s]);
            }
            const method0::ReadVerdict v = method0::flagRead(kmerIds, maxSkip, maxDrift, maxMarkerFrequency,
                alignedFractionThreshold, nearDiagonalFractionThreshold, deltaThreshold, alignment);
            flags[r] = v.palindromic ? 1 : 0;
            if(alignedCount) alignedCount[r] = v.alignedMarkerCount;
            if(nearDiagonalCount) nearDiagonalCount[r] = v.nearDiagonalMarkerCount;
            if(digests) digests[r] = method0::digest(alignment);
        }
        return 0;
    } catch(std::exception& e) { lastError = e.what(); return 1; }
}

}  // extern "C"
