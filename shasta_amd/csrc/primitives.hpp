// Device-wide building blocks written for gfx950 wave64: exclusive scan and a
// stable LSD radix sort (8-bit digits, wave-match ranking).  These replace the
// CPU structures of the reference's bucket build: the count[] / toc[] arrays of
// MemoryMapped::VectorOfVectors::beginPass1/beginPass2/storeMultithreaded
// (src/MemoryMappedVectorOfVectors.hpp:315-393) and the per-read std::sort +
// merge of LowHash0::pass3ThreadFunction (src/LowHash0.cpp:462-468).
#pragma once

#include "common.hpp"

namespace shasta_mi355x {

constexpr int WAVE = 64;

__device__ __forceinline__ int laneId() { return int(threadIdx.x) & 63; }
__device__ __forceinline__ uint64_t laneMaskLt() { return (1ULL << laneId()) - 1ULL; }

// Id of the reverse complement of a k-mer (k <= 16).  A k-mer id is two k-bit planes, (high bits of
// the k bases) << k | (low bits), first base at the plane's most significant bit
// (/root/reference/src/ShortBaseSequence.hpp:89-105); complementing flips every bit, reversing
// reverses each plane (:109-117).
__device__ __forceinline__ uint32_t reverseComplementKmerId(uint32_t kmerId, uint32_t k)
{
    const uint32_t mask = uint32_t((1ULL << k) - 1ULL);
    const uint32_t low = ~kmerId & mask, high = ~(kmerId >> k) & mask;
    return ((__brev(high) >> (32u - k)) << k) | (__brev(low) >> (32u - k));
}

// s_store_dwordx4: 16 bytes that live in scalar registers (wave-uniform by construction: ballots) go to global memory without
// passing through a vector register.  `address` must be wave-uniform (an SGPR pair) and 4-byte aligned.  The scalar data cache
// is write-back: scalarStoreFlush() before the wavefront ends, or a later kernel may not see the data.
// (The wave64 emulator of tests/emu supplies its own: SHASTA_SCALAR_STORE_DEFINED.)
#if !defined(SHASTA_SCALAR_STORE_DEFINED) && defined(SHASTA_TRACE_VECTOR_STORES)
// (the same three entry points as plain vector stores of lane 0: the A/B build of the search for round 5's method-3 failure, and the
// form to fall back on should scalar stores ever be at fault -- make OUT=../_build_vector_stores EXTRA=-DSHASTA_TRACE_VECTOR_STORES=1)
#define SHASTA_SCALAR_STORE_DEFINED
__device__ __forceinline__ void scalarStore128(void* address, uint64_t low, uint64_t high)
{
    if(laneId() == 0) *reinterpret_cast<uint4*>(address) = make_uint4(uint32_t(low), uint32_t(low >> 32), uint32_t(high), uint32_t(high >> 32));
}
__device__ __forceinline__ void scalarStore128At(void* address, int byteOffset, uint64_t low, uint64_t high)
{
    scalarStore128(static_cast<char*>(address) + byteOffset, low, high);
}
__device__ __forceinline__ void scalarStoreFlush() {}
template<class T> __device__ __forceinline__ T* uniformPointer(T* p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(v)), hi = __builtin_amdgcn_readfirstlane(uint32_t(v >> 32));
    return reinterpret_cast<T*>((uint64_t(hi) << 32) | lo);
}
#endif
#ifndef SHASTA_SCALAR_STORE_DEFINED
__device__ __forceinline__ void scalarStore128(void* address, uint64_t low, uint64_t high)
{
    const __uint128_t v = (__uint128_t(high) << 64) | __uint128_t(low);
    asm volatile("s_store_dwordx4 %0, %1, 0x0" :: "s"(v), "s"(address) : "memory");
}
// The same at a byte offset that is a constant by the time the code is generated (an expression of unrolled loop counters):
// it goes into the instruction, no address arithmetic.
__device__ __forceinline__ void scalarStore128At(void* address, int byteOffset, uint64_t low, uint64_t high)
{
    const __uint128_t v = (__uint128_t(high) << 64) | __uint128_t(low);
    asm volatile("s_store_dwordx4 %0, %1, %2" :: "s"(v), "s"(address), "i"(byteOffset) : "memory");
}
__device__ __forceinline__ void scalarStoreFlush() { asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); }
// A wave-uniform pointer the compiler cannot prove uniform (it depends on threadIdx.x >> 6).
template<class T> __device__ __forceinline__ T* uniformPointer(T* p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(v)), hi = __builtin_amdgcn_readfirstlane(uint32_t(v >> 32));
    return reinterpret_cast<T*>((uint64_t(hi) << 32) | lo);
}
#endif

// An element count a kernel is given: an upper bound known on the host (grids are sized from it) and, optionally, the
// address of the exact count on the device, written by an earlier kernel of the same stream.  The host never reads the
// exact count back between the launches: no stream synchronisation inside a chain of kernels.
struct Count {
    uint64_t bound;
    const unsigned long long* device;
    Count(uint64_t n) : bound(n), device(nullptr) {}
    Count(uint64_t n, const unsigned long long* exact) : bound(n), device(exact) {}
    __device__ __forceinline__ uint64_t get() const
    {
        if(device == nullptr) return bound;
        const uint64_t exact = uint64_t(*device);
        return exact < bound ? exact : bound;
    }
};

// ----------------------------------------------------------------------------
// Exclusive scan.
// ----------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// Block-wide exclusive scan of one value per thread; returns the exclusive
// prefix and writes the block total to *total (valid in every thread).
template<class T>
__device__ __forceinline__ T blockExclusiveScan(T v, T* total)
{
    __shared__ T waveSums[SCAN_THREADS / WAVE];
    const int lane = laneId();
    const int wave = int(threadIdx.x) >> 6;
    T inclusive = v;
#pragma unroll
    for(int d = 1; d < WAVE; d <<= 1) {
        const T o = __shfl_up(inclusive, d, WAVE);
        if(lane >= d) inclusive += o;
    }
    if(lane == WAVE - 1) waveSums[wave] = inclusive;
    __syncthreads();
    T waveOffset = 0, all = 0;
#pragma unroll
    for(int w = 0; w < SCAN_THREADS / WAVE; w++) {
        const T s = waveSums[w];
        if(w < wave) waveOffset += s;
        all += s;
    }
    __syncthreads();
    *total = all;
    return waveOffset + inclusive - v;
}

template<class T>
__global__ void __launch_bounds__(SCAN_THREADS)
scanReduceKernel(const T* __restrict__ in, T* __restrict__ blockSums, uint64_t n)
{
    const uint64_t base = uint64_t(blockIdx.x) * SCAN_TILE + uint64_t(threadIdx.x) * SCAN_ITEMS;
    T s = 0;
#pragma unroll
    for(int i = 0; i < SCAN_ITEMS; i++) if(base + i < n) s += in[base + i];
    T total;
    (void)blockExclusiveScan<T>(s, &total);
    if(threadIdx.x == 0) blockSums[blockIdx.x] = total;
}

template<class T>
__global__ void __launch_bounds__(SCAN_THREADS)
scanDownsweepKernel(const T* in, T* out, const T* blockOffsets, uint64_t n)   // out may alias in
{
    const uint64_t base = uint64_t(blockIdx.x) * SCAN_TILE + uint64_t(threadIdx.x) * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    T s = 0;
#pragma unroll
    for(int i = 0; i < SCAN_ITEMS; i++) { v[i] = (base + i < n) ? in[base + i] : T(0); s += v[i]; }
    T total;
    T prefix = blockExclusiveScan<T>(s, &total) + (blockOffsets ? blockOffsets[blockIdx.x] : T(0));
#pragma unroll
    for(int i = 0; i < SCAN_ITEMS; i++) {
        if(base + i < n) out[base + i] = prefix;
        prefix += v[i];
    }
}

// Temporary elements needed by exclusiveScan for n inputs.
inline size_t scanTempElements(uint64_t n)
{
    size_t t = 0;
    while(n > SCAN_TILE) { n = (n + SCAN_TILE - 1) / SCAN_TILE; t += n; }
    return t + 1;
}

// out may alias in.  temp must hold scanTempElements(n) elements.
template<class T>
inline void exclusiveScan(const T* in, T* out, uint64_t n, T* temp, hipStream_t stream)
{
    if(n == 0) return;
    const unsigned blocks = divUp(n, SCAN_TILE);
    if(blocks == 1) {
        hipLaunchKernelGGL(scanDownsweepKernel<T>, dim3(1), dim3(SCAN_THREADS), 0, stream, in, out, (const T*)nullptr, n);
        return;
    }
    hipLaunchKernelGGL(scanReduceKernel<T>, dim3(blocks), dim3(SCAN_THREADS), 0, stream, in, temp, n);
    exclusiveScan<T>(temp, temp, blocks, temp + blocks, stream);
    hipLaunchKernelGGL(scanDownsweepKernel<T>, dim3(blocks), dim3(SCAN_THREADS), 0, stream, in, out, (const T*)temp, n);
}

// ----------------------------------------------------------------------------
// Stable LSD radix sort, 8 bits per pass.
// A block owns RS_TILE consecutive keys; each of its 4 waves owns a contiguous
// quarter, so (wave, round, lane) order is global index order and ranking per
// wave with ballot-match keeps the pass stable.
// ----------------------------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / WAVE;
constexpr int RS_ROUNDS = 16;
constexpr int RS_PER_WAVE = RS_ROUNDS * WAVE;
constexpr int RS_TILE = RS_WAVES * RS_PER_WAVE;      // 4096 keys per block
constexpr int RS_BINS = 256;

template<class K>
__global__ void __launch_bounds__(RS_THREADS)
radixHistogramKernel(const K* __restrict__ keys, uint32_t* __restrict__ counts, Count count, int shift, unsigned numBlocks)
{
    __shared__ uint32_t hist[RS_BINS];
    const uint64_t n = count.get();
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t tileBase = uint64_t(blockIdx.x) * RS_TILE;
#pragma unroll 4
    for(int r = 0; r < RS_TILE / RS_THREADS; r++) {
        const uint64_t i = tileBase + uint64_t(r) * RS_THREADS + threadIdx.x;
        if(i < n) atomicAdd(&hist[unsigned(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    counts[uint64_t(threadIdx.x) * numBlocks + blockIdx.x] = hist[threadIdx.x];
}

// The tile leaves through LDS: every key is first put where it belongs inside the tile's own sorted order (digit by digit,
// and inside a digit in index order), then the tile is written out position by position -- neighbouring positions of one digit
// go to neighbouring addresses, so that a store instruction writes runs of whole lines where storing from the ranking's
// registers wrote 64 scattered words (round 3: the sort of 3e7 twelve-byte records 4.0 -> 2.x ms).
template<class K, class V, bool HAS_V>
__global__ void __launch_bounds__(RS_THREADS)
radixScatterKernel(const K* __restrict__ keysIn, K* __restrict__ keysOut,
    const V* __restrict__ valsIn, V* __restrict__ valsOut,
    const uint32_t* __restrict__ offsets, Count count, int shift, unsigned numBlocks)
{
    __shared__ uint32_t counters[RS_WAVES][RS_BINS];   // per-wave digit counts, then the wave's first position of the digit inside the tile
    __shared__ uint32_t tileStart[RS_BINS];            // first position of the digit inside the tile
    __shared__ uint32_t globalStart[RS_BINS];          // ... and in the output
    __shared__ uint32_t waveTotals[RS_WAVES];
    __shared__ K stagedKeys[RS_TILE];
    __shared__ V stagedVals[HAS_V ? RS_TILE : 1];
    // LDS per workgroup: the staged tile + 6 KB of tables -- 70 KB for 8-byte keys with 8-byte values (the one-pass LowHash0's
    // wide record keys and its partition by owner): two workgroups per CU of gfx950's 160 KB; beyond what a 64-KB-LDS part holds.
    static_assert(sizeof(K) * RS_TILE + (HAS_V ? sizeof(V) * RS_TILE : 0) + (RS_WAVES + 2) * RS_BINS * 4 + 64 <= 80 * 1024, "radixScatterKernel: LDS per workgroup (gfx950: 160 KB per CU)");
    const uint64_t n = count.get();
    const int lane = laneId();
    const int wave = int(threadIdx.x) >> 6;
#pragma unroll
    for(int w = 0; w < RS_WAVES; w++) counters[w][threadIdx.x] = 0;
    __syncthreads();

    const uint64_t tileBase = uint64_t(blockIdx.x) * RS_TILE;
    const uint64_t waveBase = tileBase + uint64_t(wave) * RS_PER_WAVE;
    K key[RS_ROUNDS];
    uint32_t rank[RS_ROUNDS];
#pragma unroll
    for(int r = 0; r < RS_ROUNDS; r++) {
        const uint64_t i = waveBase + uint64_t(r) * WAVE + lane;
        const bool valid = i < n;
        key[r] = valid ? keysIn[i] : K(0);
        const unsigned digit = unsigned(key[r] >> shift) & 255u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for(int b = 0; b < 8; b++) {
            const uint64_t vote = __ballot((digit >> b) & 1u);
            peers &= ((digit >> b) & 1u) ? vote : ~vote;
        }
        // Wave-synchronous: one wave owns counters[wave][*].
        uint32_t prev = 0;
        const int leader = __ffsll((unsigned long long)peers) - 1;
        if(valid && lane == leader) {
            prev = counters[wave][digit];
            counters[wave][digit] = prev + uint32_t(__popcll(peers));
        }
        prev = __shfl(prev, leader < 0 ? 0 : leader, WAVE);
        rank[r] = prev + uint32_t(__popcll(peers & laneMaskLt()));
        __builtin_amdgcn_wave_barrier();   // keep the per-wave LDS read-modify-write of successive rounds in order
    }
    __syncthreads();
    {
        // Thread d: the tile's count of digit d, its exclusive scan over the digits (a wave scan + the waves' totals), the first
        // position of every wave's keys of digit d inside the tile, and where the digit starts in the output.
        const unsigned d = threadIdx.x;
        uint32_t c[RS_WAVES], total = 0;
#pragma unroll
        for(int w = 0; w < RS_WAVES; w++) { c[w] = counters[w][d]; total += c[w]; }
        uint32_t inclusive = total;
#pragma unroll
        for(int delta = 1; delta < WAVE; delta <<= 1) { const uint32_t o = uint32_t(__shfl_up(int(inclusive), delta, WAVE)); if(lane >= delta) inclusive += o; }
        if(lane == WAVE - 1) waveTotals[wave] = inclusive;
        __syncthreads();
        uint32_t start = inclusive - total;
        for(int w = 0; w < wave; w++) start += waveTotals[w];
        tileStart[d] = start;
        globalStart[d] = offsets[uint64_t(d) * numBlocks + blockIdx.x];
#pragma unroll
        for(int w = 0; w < RS_WAVES; w++) { counters[w][d] = start; start += c[w]; }
    }
    __syncthreads();
#pragma unroll
    for(int r = 0; r < RS_ROUNDS; r++) {
        const uint64_t i = waveBase + uint64_t(r) * WAVE + lane;
        if(i < n) {
            const unsigned digit = unsigned(key[r] >> shift) & 255u;
            const uint32_t at = counters[wave][digit] + rank[r];
            stagedKeys[at] = key[r];
            if(HAS_V) stagedVals[at] = valsIn[i];
        }
    }
    __syncthreads();
    const uint32_t inTile = uint32_t(n > tileBase ? (n - tileBase < uint64_t(RS_TILE) ? n - tileBase : uint64_t(RS_TILE)) : 0);
#pragma unroll 4
    for(int r = 0; r < RS_TILE / RS_THREADS; r++) {
        const uint32_t at = uint32_t(r) * RS_THREADS + threadIdx.x;
        if(at < inTile) {
            const K k = stagedKeys[at];
            const unsigned digit = unsigned(k >> shift) & 255u;
            const uint32_t dst = globalStart[digit] + (at - tileStart[digit]);
            keysOut[dst] = k;
            if(HAS_V) valsOut[dst] = stagedVals[at];
        }
    }
}

struct RadixSortWorkspace {
    DeviceBuffer<uint32_t> counts;
    DeviceBuffer<uint32_t> scanTemp;
};

// Sorts n (< 2^32) keys on bits [firstBit, firstBit + bits) with 8-bit passes, ping-ponging between
// (keysA, valsA) and (keysB, valsB).  Returns true if the result is in B (which side holds the result depends on
// `bits` only, not on the count).
template<class K, class V, bool HAS_V>
inline bool radixSort(K* keysA, K* keysB, V* valsA, V* valsB, Count count, int bits,
    RadixSortWorkspace& ws, hipStream_t stream, int firstBit = 0)
{
    const uint64_t n = count.bound;                    // grids and workspace from the bound; the kernels use the exact count
    if(n == 0 || bits <= 0) return false;
    MI355X_ASSERT(n < (1ULL << 32));
    const unsigned numBlocks = divUp(n, RS_TILE);
    const uint64_t countN = uint64_t(RS_BINS) * numBlocks;
    ws.counts.reserve(countN, stream);
    ws.scanTemp.reserve(scanTempElements(countN), stream);
    bool inB = false;
    for(int shift = firstBit; shift < firstBit + bits; shift += 8) {
        K* kin = inB ? keysB : keysA;  K* kout = inB ? keysA : keysB;
        V* vin = inB ? valsB : valsA;  V* vout = inB ? valsA : valsB;
        hipLaunchKernelGGL(radixHistogramKernel<K>, dim3(numBlocks), dim3(RS_THREADS), 0, stream,
            (const K*)kin, ws.counts.data(), count, shift, numBlocks);
        exclusiveScan<uint32_t>(ws.counts.data(), ws.counts.data(), countN, ws.scanTemp.data(), stream);
        hipLaunchKernelGGL((radixScatterKernel<K, V, HAS_V>), dim3(numBlocks), dim3(RS_THREADS), 0, stream,
            (const K*)kin, kout, (const V*)vin, vout, (const uint32_t*)ws.counts.data(), count, shift, numBlocks);
        inB = !inB;
    }
    return inB;
}

}  // namespace shasta_mi355x
