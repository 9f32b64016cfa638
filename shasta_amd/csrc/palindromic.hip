// Palindromic-read screening on MI355X (gfx950): the device half of Assembler::flagPalindromicReads
// (/root/reference/src/AssemblerAlign.cpp:652-770).
//
// The reference aligns every read with its own reverse complement by method 0 (src/AlignmentGraph.cpp:
// two unstable std::sort calls and a binary-heap shortest path whose tie order is the C++ library's)
// and flags the read when the alignment has enough markers and enough of them lie within
// deltaThreshold of the diagonal (:738-752).  Every aligned marker pair is a pair of equal kmer ids
// (i on strand 0, j on strand 1) (src/AlignmentGraph.cpp:214-235), so
//
//     bound(read) = #{ (i, j) : kmerId0[i] == kmerId1[j], |i - j| < deltaThreshold }
//
// is an upper bound on nearDiagonalMarkerCount whatever path the search takes, and a read with
// double(bound) / double(n) < nearDiagonalFractionThreshold cannot be flagged.  That is exact, and it
// settles all but the few reads that really are palindromic (or low-complexity): those go to the
// host, which runs the reference's sequential graph search on them (shasta_amd/host/PalindromicReads.cpp).
//
//   palindromicScreenKernel  one wavefront per read.  A tile = 64 consecutive i, one per lane; the
//                            j they can pair with, [i0 - delta + 1, i0 + 62 + delta], sit in the
//                            wavefront's LDS window (strand 1 is read once per tile, coalesced); each
//                            lane then walks its 2 delta - 1 offsets: consecutive lanes read
//                            consecutive LDS words, no bank conflicts.
// Work: n (2 delta - 1) compares per read, 4 n (1 + (64 + 2 delta) / 64) bytes from HBM/L2.
#include "context.hpp"

#include <cstring>
#include <mutex>

namespace shasta_mi355x {
namespace {

constexpr int SCREEN_WAVES = 4;               // wavefronts (reads) per workgroup

__global__ void __launch_bounds__(64 * SCREEN_WAVES)
palindromicScreenKernel(const uint32_t* __restrict__ kmerIds, const uint64_t* __restrict__ toc, uint64_t readCount,
    uint32_t delta, uint32_t* __restrict__ bound)
{
    extern __shared__ uint32_t screenWindows[];                    // SCREEN_WAVES x (64 + 2 delta) words
    const uint32_t windowWords = 64u + 2u * delta;
    uint32_t* const window = screenWindows + (threadIdx.x >> 6) * windowWords;
    const int lane = laneId();
    const uint64_t r = uint64_t(blockIdx.x) * SCREEN_WAVES + (threadIdx.x >> 6);
    if(r >= readCount) return;                                     // whole wavefront; no block barriers below
    const uint64_t begin0 = toc[2 * r], begin1 = toc[2 * r + 1], end1 = toc[2 * r + 2];
    const uint32_t n = uint32_t(begin1 - begin0);
    const uint32_t n1 = uint32_t(end1 - begin1);                   // == n for markers made by MarkerFinder
    const uint32_t* __restrict__ a = kmerIds + begin0;
    const uint32_t* __restrict__ b = kmerIds + begin1;
    uint32_t count = 0;
    for(uint32_t i0 = 0; i0 < n; i0 += 64) {
        // window[w] = b[i0 - (delta - 1) + w]; positions outside strand 1 hold a value no kmer id has.
        const int64_t first = int64_t(i0) - int64_t(delta - 1);
        for(uint32_t w = uint32_t(lane); w < windowWords; w += 64) {
            const int64_t j = first + int64_t(w);
            window[w] = (j >= 0 && j < int64_t(n1)) ? b[j] : 0xffffffffu;
        }
        // One wavefront's LDS operations run in order; the fence drains them and stops the compiler from moving the reads up.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint32_t i = i0 + uint32_t(lane);
        if(i < n) {
            const uint32_t mine = a[i];
            // j = i + d, d = -(delta-1) .. delta-1  <=>  w = lane + (d + delta - 1) = lane .. lane + 2 delta - 2
            for(uint32_t w = uint32_t(lane); w <= uint32_t(lane) + 2u * delta - 2u; w++) count += (window[w] == mine) ? 1u : 0u;
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for(int d = 32; d >= 1; d >>= 1) count += __shfl_down(count, d, WAVE);
    if(lane == 0) bound[r] = count;
}

}  // namespace

// bound[r] for every read of the context's markers (host pointer, readCount entries).
void palindromicScreen(Context& ctx, uint64_t deltaThreshold, uint32_t* bound)
{
    if(deltaThreshold == 0 || deltaThreshold > 4096) throw std::runtime_error("palindromic_screen: deltaThreshold must be in [1, 4096] (LDS window of 64 + 2 delta words per wavefront).");
    const uint64_t readCount = ctx.readCount;
    if(readCount == 0) return;
    hipStream_t stream = ctx.stream;
    DeviceBuffer<uint32_t> d;
    d.reserve(readCount, stream);
    const size_t ldsBytes = size_t(SCREEN_WAVES) * (64 + 2 * size_t(deltaThreshold)) * sizeof(uint32_t);
    if(ldsBytes > 48 * 1024) {
        // Above the default limit of dynamic LDS per workgroup (the largest window is 132 KB of gfx950's 160 KB).
        std::call_once(ctx.palindromicLdsAttribute, [] {                    // per context = per device
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&palindromicScreenKernel),
                hipFuncAttributeMaxDynamicSharedMemorySize, int(size_t(SCREEN_WAVES) * (64 + 2 * 4096) * sizeof(uint32_t))));
        });
    }
    hipLaunchKernelGGL(palindromicScreenKernel, dim3(divUp(readCount, SCREEN_WAVES)), dim3(64 * SCREEN_WAVES), ldsBytes, stream,
        (const uint32_t*)ctx.kmerIds.data(), (const uint64_t*)ctx.toc.data(), readCount, uint32_t(deltaThreshold), d.data());
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(bound, d.data(), readCount * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
}

}  // namespace shasta_mi355x
