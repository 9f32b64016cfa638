// Marker finding on MI355X (gfx950): the producer of the hot path's input.
// Replaces MarkerFinder::MarkerFinder (/root/reference/src/MarkerFinder.cpp:16-127), called by
// Assembler::findMarkers (src/AssemblerMarkers.cpp:11-24): for every read, every k-mer that the
// k-mer table flags as a marker becomes a marker of strand 0 (k-mer id, position) and, reverse
// complemented, of strand 1 (id of the reverse complement, baseCount - k - position; strand 1 is
// stored in its own position order, i.e. backwards, :96-100).
//
// Reads arrive as Shasta stores them (LongBaseSequences, src/LongBaseSequence.hpp:33-41): per block
// of 64 bases two 64-bit words, the low bits of the bases then the high bits, base 0 in the most
// significant bit.  A k-mer id is (high bits of its k bases) << k | (low bits), first base most
// significant (src/ShortBaseSequence.hpp:89-105) -- so the id at position p is two k-bit fields cut
// out of the two bit planes: no rolling state, every position is independent.
//
//   markerCountKernel  one wavefront per read, 64 positions per round: k-mer id, one bit of the
//                      marker bitmap (4^k bits: 128 KB for k = 10, L2-resident), __ballot count;
//   markerWriteKernel  the same sweep with ballot compaction: dense kmerIds[] / positions[] of both
//                      strands at the offsets of the scanned counts;
//   packMarkersKernel  the 7-byte CompressedMarker records (src/Marker.hpp:56-70) for the caller
//                      that wants Markers.data back.
// HBM-bound streaming: 0.25 B read per base, 8 B written per marker per strand (+7 B packed).
// With a context the dense kmer ids stay resident: LowHash0 / the aligners run on them with no
// 7-byte-per-marker upload (the 2-bit reads are a quarter of the size of their markers).
#include "context.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>

namespace shasta_mi355x {
namespace {

// The k-mer id at base position p of a read (p + k <= baseCount, k <= 16).
__device__ __forceinline__ uint32_t kmerIdAt(const uint64_t* __restrict__ words, uint64_t p, uint32_t k)
{
    const uint64_t block = p >> 6;
    const uint32_t o = uint32_t(p & 63);
    uint64_t low = words[2 * block] << o, high = words[2 * block + 1] << o;
    if(o + k > 64) {                                   // the k-mer straddles two blocks; o >= 49 here
        low |= words[2 * block + 2] >> (64 - o);
        high |= words[2 * block + 3] >> (64 - o);
    }
    return (uint32_t(high >> (64 - k)) << k) | uint32_t(low >> (64 - k));
}

template<bool WRITE>
__global__ void __launch_bounds__(256)
markerSweepKernel(const uint64_t* __restrict__ readsData, const uint64_t* __restrict__ readsToc, const uint64_t* __restrict__ baseCounts,
    uint64_t readCount, uint32_t k, const uint32_t* __restrict__ markerBitmap,
    uint64_t* __restrict__ counts,                                              // !WRITE: [2R+1] markers per oriented read
    const uint64_t* __restrict__ toc, uint32_t* __restrict__ kmerIds, uint32_t* __restrict__ positions)
{
    const uint64_t r = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int lane = laneId();
    if(r > readCount) return;
    if(r == readCount) {
        if(!WRITE && lane == 0) counts[2 * r] = 0;
        return;
    }
    const uint64_t* __restrict__ words = readsData + readsToc[r];
    const uint64_t baseCount = baseCounts[r];
    const uint64_t positionCount = baseCount >= k ? baseCount - k + 1 : 0;     // :78
    const uint64_t begin0 = WRITE ? toc[2 * r] : 0, end1 = WRITE ? toc[2 * r + 2] : 0;
    uint64_t found = 0;                                                        // wave-uniform
    for(uint64_t base = 0; base < positionCount; base += WAVE) {
        const uint64_t p = base + uint64_t(lane);
        uint32_t id = 0;
        bool isMarker = false;
        if(p < positionCount) {
            id = kmerIdAt(words, p, k);
            isMarker = (markerBitmap[id >> 5] >> (id & 31u)) & 1u;
        }
        const uint64_t votes = __ballot(isMarker);
        if(WRITE && isMarker) {
            const uint64_t rank = found + uint64_t(__popcll(votes & laneMaskLt()));
            kmerIds[begin0 + rank] = id;
            positions[begin0 + rank] = uint32_t(p);
            kmerIds[end1 - 1 - rank] = reverseComplementKmerId(id, k);
            positions[end1 - 1 - rank] = uint32_t(baseCount - k - p);
        }
        found += uint64_t(__popcll(votes));
    }
    if(!WRITE && lane == 0) { counts[2 * r] = found; counts[2 * r + 1] = found; }
}

// CompressedMarker: 4 bytes of k-mer id, 3 bytes of position, packed (src/Marker.hpp:56-70).
__global__ void __launch_bounds__(256)
packMarkersKernel(const uint32_t* __restrict__ kmerIds, const uint32_t* __restrict__ positions, uint64_t markerCount, uint8_t* __restrict__ data7)
{
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    for(uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < markerCount; i += stride) {
        const uint32_t id = kmerIds[i], position = positions[i];
        uint8_t* out = data7 + 7 * i;
        out[0] = uint8_t(id); out[1] = uint8_t(id >> 8); out[2] = uint8_t(id >> 16); out[3] = uint8_t(id >> 24);
        out[4] = uint8_t(position); out[5] = uint8_t(position >> 8); out[6] = uint8_t(position >> 16);
    }
}

}  // namespace

void findMarkers(Context& ctx, uint64_t readCount, const uint64_t* readsToc, const uint64_t* readsData, const uint64_t* baseCounts,
    uint64_t k, const void* kmerTable, uint64_t kmerInfoStride, uint64_t isMarkerOffset, const uint8_t* readFlags,
    bool wantPacked, shasta_markers_result& result)
{
    std::memset(&result, 0, sizeof(result));
    const auto t0 = std::chrono::steady_clock::now();
    if(k < 1 || k > 16) throw std::runtime_error("find_markers: k must be in [1, 16].");
    if(kmerInfoStride == 0 || isMarkerOffset >= kmerInfoStride) throw std::runtime_error("find_markers: bad k-mer table layout.");
    MI355X_ASSERT(readCount < (1ULL << 31));
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;

    // The isMarker flags as one bit per k-mer id.
    const uint64_t kmerCount = 1ULL << (2 * k);
    std::vector<uint32_t> bitmap((kmerCount + 31) / 32, 0u);
    const uint8_t* table = static_cast<const uint8_t*>(kmerTable);
    for(uint64_t id = 0; id < kmerCount; id++) if(table[id * kmerInfoStride + isMarkerOffset]) bitmap[id >> 5] |= 1u << (id & 31);

    for(uint64_t r = 0; r < readCount; r++) {
        MI355X_ASSERT(readsToc[r] <= readsToc[r + 1]);
        MI355X_ASSERT(baseCounts[r] < (1ULL << 24));                                   // Uint24 positions
        MI355X_ASSERT(readsToc[r + 1] - readsToc[r] >= (baseCounts[r] ? 2 * ((baseCounts[r] - 1) / 64 + 1) : 0));
    }
    const uint64_t wordCount = readsToc[readCount];
    DeviceBuffer<uint64_t> dData, dToc, dBaseCounts, dMarkersToc, scanTemp;
    DeviceBuffer<uint32_t> dBitmap, dKmerIds, dPositions;
    dData.reserve(wordCount + 4, stream); dToc.reserve(readCount + 1, stream); dBaseCounts.reserve(readCount + 1, stream);
    dBitmap.reserve(bitmap.size(), stream); dMarkersToc.reserve(2 * readCount + 1, stream);
    scanTemp.reserve(scanTempElements(2 * readCount + 1), stream);
    hipEvent_t evBegin, evEnd;
    HIP_CHECK(hipEventCreate(&evBegin)); HIP_CHECK(hipEventCreate(&evEnd));
    HIP_CHECK(hipEventRecord(evBegin, stream));
    HIP_CHECK(hipMemsetAsync(dData.data() + wordCount, 0, 4 * sizeof(uint64_t), stream));
    if(wordCount) HIP_CHECK(hipMemcpyAsync(dData.data(), readsData, wordCount * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(dToc.data(), readsToc, (readCount + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    if(readCount) HIP_CHECK(hipMemcpyAsync(dBaseCounts.data(), baseCounts, readCount * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(dBitmap.data(), bitmap.data(), bitmap.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));

    const unsigned grid = divUp((readCount + 1) * WAVE, 256);
    // Algorithmic bytes: 0.25 B per base read by either sweep (two bit planes); the second also writes 8 B per marker and strand.
    uint64_t baseCount = 0;
    for(uint64_t r = 0; r < readCount; r++) baseCount += baseCounts[r];
    SHASTA_TIMED(ctx, "markerSweepKernel<false> (count)", stream, baseCount / 4, baseCount,
        hipLaunchKernelGGL(markerSweepKernel<false>, dim3(grid), dim3(256), 0, stream,
            (const uint64_t*)dData.data(), (const uint64_t*)dToc.data(), (const uint64_t*)dBaseCounts.data(), readCount, uint32_t(k),
            (const uint32_t*)dBitmap.data(), dMarkersToc.data(), (const uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr));
    HIP_CHECK(hipGetLastError());
    exclusiveScan<uint64_t>(dMarkersToc.data(), dMarkersToc.data(), 2 * readCount + 1, scanTemp.data(), stream);
    std::vector<uint64_t> markersToc(2 * readCount + 1);
    HIP_CHECK(hipMemcpyAsync(markersToc.data(), dMarkersToc.data(), markersToc.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    const uint64_t markerCount = markersToc[2 * readCount];
    dKmerIds.reserve(markerCount + 1, stream); dPositions.reserve(markerCount + 1, stream);
    SHASTA_TIMED(ctx, "markerSweepKernel<true> (write)", stream, baseCount / 4 + 8 * markerCount, baseCount,
        hipLaunchKernelGGL(markerSweepKernel<true>, dim3(grid), dim3(256), 0, stream,
            (const uint64_t*)dData.data(), (const uint64_t*)dToc.data(), (const uint64_t*)dBaseCounts.data(), readCount, uint32_t(k),
            (const uint32_t*)dBitmap.data(), (uint64_t*)nullptr, (const uint64_t*)dMarkersToc.data(), dKmerIds.data(), dPositions.data()));
    HIP_CHECK(hipGetLastError());

    result.markerCount = markerCount;
    result.markersToc = static_cast<uint64_t*>(std::malloc(markersToc.size() * sizeof(uint64_t)));
    if(!result.markersToc) throw std::bad_alloc();
    std::memcpy(result.markersToc, markersToc.data(), markersToc.size() * sizeof(uint64_t));
    if(wantPacked) {
        DeviceBuffer<uint8_t> dPacked;
        dPacked.reserve(7 * markerCount + 1, stream);
        if(markerCount) {
            const unsigned blocks = unsigned(std::min<uint64_t>(divUp(markerCount, 256), 16384));
            SHASTA_TIMED(ctx, "packMarkersKernel", stream, 15 * markerCount, markerCount,
                hipLaunchKernelGGL(packMarkersKernel, dim3(blocks), dim3(256), 0, stream,
                    (const uint32_t*)dKmerIds.data(), (const uint32_t*)dPositions.data(), markerCount, dPacked.data()));
            HIP_CHECK(hipGetLastError());
        }
        result.markersData = static_cast<uint8_t*>(std::malloc(std::max<uint64_t>(1, 7 * markerCount)));
        if(!result.markersData) throw std::bad_alloc();
        if(markerCount) HIP_CHECK(hipMemcpyAsync(result.markersData, dPacked.data(), 7 * markerCount, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    HIP_CHECK(hipEventRecord(evEnd, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, evBegin, evEnd));
    result.deviceSeconds = ms * 1e-3;
    (void)hipEventDestroy(evBegin); (void)hipEventDestroy(evEnd);

    // The context now holds these markers exactly as after set_markers: dense kmer ids + toc in HBM.
    ctx.setMarkers(readCount, markersToc.data(), nullptr, dKmerIds.data(), readFlags, true);
    result.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void findMarkersFree(shasta_markers_result& r)
{
    std::free(r.markersToc); std::free(r.markersData);
    std::memset(&r, 0, sizeof(r));
}

}  // namespace shasta_mi355x
