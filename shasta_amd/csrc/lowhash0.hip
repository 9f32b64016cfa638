// LowHash0 on MI355X (gfx950).  Replaces LowHash0::LowHash0 and its passes
// (/root/reference/src/LowHash0.cpp:23-257, :261-308, :314-484, :493-613).
//
// The reference keeps 2^log2 bucket counters and fills a CSR of buckets with
// atomics; here every iteration is
//   K1  hashWindowsKernel    one pass over the dense kmerIds (4 B/marker), MurmurHash64A per
//                            m-marker window, keep hash < threshold, block-staged compaction
//   K2  radix sort of the ~f*M records on the bucket id (replaces count/toc/fill)
//   K3  bucket boundaries, per-read {sparse,good,crowded} statistics, bucket-size histogram
//   K4  all ordered pairs of each admissible bucket -> 64-bit pair keys
//   K5  sort + run-length of the pair keys, merge into the running (key, frequency) table
//   K6  (after the loop) frequency >= minFrequency -> OrientedReadPair list
// Bit-exactness notes follow SURVEY.md Appendix A.1.
#include "context.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>

namespace shasta_mi355x {

// ---------------------------------------------------------------------------
// MurmurHash64A (src/MurmurHash2.cpp:96-140) specialised for a window of m
// little-endian uint32 kmer ids: len = 4m, blocks = pairs of ids, odd m leaves
// a 4-byte tail (switch cases 4..1 = the 32-bit word itself).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t murmurMix(uint64_t k)
{
    const uint64_t mul = 0xc6a4a7935bd1e995ULL;
    k *= mul; k ^= k >> 47; k *= mul;
    return k;
}

template<int M_FIXED>
__device__ __forceinline__ uint64_t murmurWindow(const uint32_t* w, uint32_t m, uint64_t seed)
{
    const uint64_t mul = 0xc6a4a7935bd1e995ULL;
    const uint32_t mm = M_FIXED ? uint32_t(M_FIXED) : m;
    uint64_t h = seed ^ (uint64_t(4u * mm) * mul);
    const uint32_t blocks = mm >> 1;
#pragma unroll
    for(uint32_t b = 0; b < blocks; b++) {
        const uint64_t k = uint64_t(w[2 * b]) | (uint64_t(w[2 * b + 1]) << 32);
        h ^= murmurMix(k);
        h *= mul;
    }
    if(mm & 1u) {
        h ^= uint64_t(w[mm - 1]);
        h *= mul;
    }
    h ^= h >> 47; h *= mul; h ^= h >> 47;
    return h;
}

// ---------------------------------------------------------------------------
// K0: CompressedMarker (packed 7 bytes: u32 kmerId, u24 position,
// src/Marker.hpp:56-70) -> dense kmerIds.  LowHash0::createKmerIds :286-308.
// words must be readable 8 bytes past the last marker.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
stripMarkersKernel(const uint32_t* __restrict__ words, uint32_t* __restrict__ kmerIds, uint64_t n)
{
    for(uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x) {
        const uint64_t byte = 7ULL * i;
        const uint64_t w = byte >> 2;
        const uint32_t lo = words[w], hi = words[w + 1];
        kmerIds[i] = __builtin_amdgcn_alignbyte(hi, lo, uint32_t(byte & 3u));
    }
}

// Per hash tile (256 consecutive markers): the oriented read that owns its first marker, where
// that read ends, and the read's palindromic flag -- everything most threads of the tile need,
// in one 16-byte record (upper_bound on toc).
__global__ void __launch_bounds__(256)
tileDescKernel(const uint64_t* __restrict__ toc, const uint8_t* __restrict__ readFlags, uint64_t orientedReadCount, uint64_t markerCount,
    uint4* __restrict__ tileDesc, uint64_t tileCount)
{
    const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(t > tileCount) return;
    const uint64_t i = t * HASH_TILE;
    uint32_t r = uint32_t(orientedReadCount ? orientedReadCount - 1 : 0);
    if(i < markerCount) {
        uint64_t lo = 0, hi = orientedReadCount + 1;       // first idx in [0, 2R] with toc[idx] > i
        while(lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if(toc[mid] > i) hi = mid; else lo = mid + 1;
        }
        r = uint32_t(lo - 1);
    }
    const uint64_t end = orientedReadCount ? toc[r + 1] : 0;
    const uint32_t flags = orientedReadCount ? uint32_t(readFlags[r >> 1] & 1u) : 0u;
    tileDesc[t] = make_uint4(r, flags, uint32_t(end), uint32_t(end >> 32));
}

// ---------------------------------------------------------------------------
// K1: pass 1 of the reference (src/LowHash0.cpp:314-360).
// What bounds it is the vector ALU, not HBM: MurmurHash64A costs five 64 x 64-bit multiplies per window at m = 4
// even with everything shared that can be (profiles/r02_valu_rates.jsonl: a v_mul_lo_u32 issues at 0.6 of the rate
// of an add), about 45 VALU instructions per window against 4 bytes read.  The round-1 kernel (one window per
// thread, 256-marker block tiles in LDS, three block barriers per tile) issued 103 per window and ran at 0.90 ms per
// launch of 3.0e8 windows whichever multiplies were shared.  This one has no block-level step at all:
//  * a wavefront owns tiles of HASH_TILE = 252 consecutive markers: lane l < 63 hashes the FOUR windows that start at
//    markers 4l .. 4l+3 of the tile, lane 63 only feeds its neighbour (its markers are the first four of the next tile);
//  * one 16-byte load per lane and tile, issued one tile ahead; the markers and block transforms a lane needs from the
//    next lane arrive by DPP wave shifts (m = 3, 4, 5: at most four markers and two transforms);
//  * murmurMix of the 8-byte block that starts at a marker is computed once, by the lane that owns the marker
//    (a block is block b of window j - 2b): 2 + m/2 + (m odd) + 1 multiplies per window instead of 3 (m/2) + (m odd) + 1;
//  * low hashes go to a wavefront-private LDS stage whose fill level lives in a scalar register: no LDS atomics, one
//    global atomic per ~190 records.
// Any other m runs the same kernel with window contents loaded marker by marker (L1-resident re-reads): correct
// for every m <= 33, not tuned.
// ---------------------------------------------------------------------------
constexpr int HASH_THREADS = 256;
constexpr int HASH_HALO = 32;                 // supports m <= 33
constexpr int HASH_STAGE = 512;               // records staged per wavefront (a tile adds at most HASH_TILE)

// Lane l gets the value of lane l + 1 (wave_shl:1); lane 63 gets 0.
__device__ __forceinline__ uint32_t fromNextLane(uint32_t v)
{
    return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ uint64_t fromNextLane64(uint64_t v)
{
    return uint64_t(fromNextLane(uint32_t(v))) | (uint64_t(fromNextLane(uint32_t(v >> 32))) << 32);
}

template<int M_FIXED>
__global__ void __launch_bounds__(HASH_THREADS)
hashWindowsKernel(
    const uint32_t* __restrict__ kmerIds, const uint64_t* __restrict__ toc,
    const uint8_t* __restrict__ readFlags, const uint4* __restrict__ tileDesc,
    uint64_t markerBegin, uint64_t markerEnd, uint64_t markerCount,
    uint32_t m, uint64_t seed, uint64_t hashThreshold, uint32_t mask,
    uint32_t* __restrict__ outKeys, uint64_t* __restrict__ outVals,
    unsigned long long* __restrict__ counter, uint64_t capacity)
{
    __shared__ uint32_t stageKeys[HASH_THREADS / WAVE][HASH_STAGE];
    __shared__ uint64_t stageVals[HASH_THREADS / WAVE][HASH_STAGE];
    constexpr uint64_t mul = 0xc6a4a7935bd1e995ULL;
    const uint32_t mm = M_FIXED ? uint32_t(M_FIXED) : m;
    const int lane = laneId();
    const uint32_t waveInBlock = threadIdx.x >> 6;
    uint32_t* const sKeys = stageKeys[waveInBlock];
    uint64_t* const sVals = stageVals[waveInBlock];
    uint32_t fill = 0;                                     // wave-uniform

    auto flush = [&]() {
        unsigned long long base = 0;
        if(lane == 0) base = atomicAdd(counter, (unsigned long long)fill);
        base = (unsigned long long)(__builtin_amdgcn_readfirstlane(uint32_t(base))) |
            ((unsigned long long)(__builtin_amdgcn_readfirstlane(uint32_t(base >> 32))) << 32);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for(uint32_t k = uint32_t(lane); k < fill; k += WAVE) {
            const uint64_t dst = base + k;
            if(dst < capacity) { outKeys[dst] = sKeys[k]; outVals[dst] = sVals[k]; }
        }
        __builtin_amdgcn_wave_barrier();
        fill = 0;
    };
    // Appends the hits of one window slot of the wavefront (at most 64) to the stage.
    auto append = [&](bool hit, uint64_t hash, uint32_t orientedReadId) {
        const uint64_t votes = __ballot(hit);
        if(votes == 0) return;
        if(hit) {
            const uint32_t slot = fill + uint32_t(__popcll(votes & laneMaskLt()));
            sKeys[slot] = uint32_t(hash) & mask;                              // bucket id, :352
            sVals[slot] = (hash & 0xffffffff00000000ULL) | orientedReadId;    // BucketEntry: hashHighBits, orientedReadId
        }
        fill += uint32_t(__popcll(votes));
    };

    const uint64_t firstTile = markerBegin / HASH_TILE;
    const uint64_t lastTile = (markerEnd + HASH_TILE - 1) / HASH_TILE;          // exclusive
    const uint64_t waves = uint64_t(gridDim.x) * (HASH_THREADS / WAVE);
    uint64_t tile = firstTile + uint64_t(blockIdx.x) * (HASH_THREADS / WAVE) + waveInBlock;
    // kmerIds is readable HASH_TILE + HASH_HALO markers past markerCount (Context::setMarkers).
    auto loadTile = [&](uint64_t t, uint4& k, uint4& desc) {
        k = *reinterpret_cast<const uint4*>(kmerIds + t * HASH_TILE + 4 * uint64_t(lane));
        desc = tileDesc[t];
    };
    uint4 nextK = make_uint4(0, 0, 0, 0), nextDesc = make_uint4(0, 0, 0, 0);
    if(tile < lastTile) loadTile(tile, nextK, nextDesc);
    for(; tile < lastTile; tile += waves) {
        const uint4 K = nextK, desc = nextDesc;
        if(tile + waves < lastTile) loadTile(tile + waves, nextK, nextDesc);
        if(fill + HASH_TILE > HASH_STAGE) flush();                            // room for every window of this tile
        const uint64_t i0 = tile * HASH_TILE + 4 * uint64_t(lane);           // first of this lane's four markers
        const bool owner = lane < WAVE - 1;                                    // lane 63 only feeds lane 62

        // The read of the lane's markers: the tile's first read unless a read boundary lies before them.
        uint32_t r = desc.x;
        uint64_t end = uint64_t(desc.z) | (uint64_t(desc.w) << 32);
        bool palindromic = (desc.y & 1u) != 0;
        // Fast path: all four windows lie in that read.  Otherwise every window finds its own read (rare: one or two
        // lanes per read boundary).
        const bool simple = i0 + 3 + mm <= end;
        auto readOf = [&](uint64_t i, uint32_t& rr, uint64_t& ee, bool& pp) {
            rr = r; ee = end; pp = palindromic;
            if(i >= ee) {
                ++rr;
                while(toc[rr + 1] <= i) ++rr;
                ee = toc[rr + 1];
                pp = (readFlags[rr >> 1] & 1u) != 0;
            }
        };

        uint64_t hash[4];
        if constexpr (M_FIXED >= 3 && M_FIXED <= 5) {
            // Markers 0..3 are the lane's own, 4..7 the next lane's.
            const uint32_t k0 = K.x, k1 = K.y, k2 = K.z, k3 = K.w;
            const uint32_t k4 = fromNextLane(K.x), k5 = fromNextLane(K.y), k6 = fromNextLane(K.z), k7 = fromNextLane(K.w);
            // Block transforms that start at the lane's own markers, and the next lane's first two.
            const uint64_t p0 = murmurMix(uint64_t(k0) | (uint64_t(k1) << 32)), p1 = murmurMix(uint64_t(k1) | (uint64_t(k2) << 32));
            const uint64_t p2 = murmurMix(uint64_t(k2) | (uint64_t(k3) << 32)), p3 = murmurMix(uint64_t(k3) | (uint64_t(k4) << 32));
            const uint64_t h0 = seed ^ (uint64_t(4u * M_FIXED) * mul);
            auto finish = [&](uint64_t h) { h ^= h >> 47; h *= mul; h ^= h >> 47; return h; };
            if constexpr (M_FIXED == 3) {
                (void)k6; (void)k7;
                hash[0] = finish(((h0 ^ p0) * mul ^ uint64_t(k2)) * mul);
                hash[1] = finish(((h0 ^ p1) * mul ^ uint64_t(k3)) * mul);
                hash[2] = finish(((h0 ^ p2) * mul ^ uint64_t(k4)) * mul);
                hash[3] = finish(((h0 ^ p3) * mul ^ uint64_t(k5)) * mul);
            } else {
                const uint64_t p4 = fromNextLane64(p0), p5 = fromNextLane64(p1);
                if constexpr (M_FIXED == 4) {
                    (void)k5; (void)k6; (void)k7;
                    hash[0] = finish(((h0 ^ p0) * mul ^ p2) * mul);
                    hash[1] = finish(((h0 ^ p1) * mul ^ p3) * mul);
                    hash[2] = finish(((h0 ^ p2) * mul ^ p4) * mul);
                    hash[3] = finish(((h0 ^ p3) * mul ^ p5) * mul);
                } else {
                    (void)k5;
                    hash[0] = finish((((h0 ^ p0) * mul ^ p2) * mul ^ uint64_t(k4)) * mul);
                    hash[1] = finish((((h0 ^ p1) * mul ^ p3) * mul ^ uint64_t(k5)) * mul);
                    hash[2] = finish((((h0 ^ p2) * mul ^ p4) * mul ^ uint64_t(k6)) * mul);
                    hash[3] = finish((((h0 ^ p3) * mul ^ p5) * mul ^ uint64_t(k7)) * mul);
                }
            }
        } else {
            // Any m: the window's markers one by one (they are in L1 after the tile load).
#pragma unroll
            for(int w = 0; w < 4; w++) {
                uint32_t win[HASH_HALO + 1];
                for(uint32_t q = 0; q < mm; q++) win[q] = kmerIds[i0 + uint32_t(w) + q];
                hash[w] = murmurWindow<0>(win, mm, seed);
            }
        }
#pragma unroll
        for(int w = 0; w < 4; w++) {
            const uint64_t i = i0 + uint32_t(w);
            bool ok = owner && i >= markerBegin && i < markerEnd;
            uint32_t rr = r;
            if(simple) ok = ok && !palindromic;
            else if(ok) {
                uint64_t ee; bool pp;
                readOf(i, rr, ee, pp);
                // Reads with fewer than m markers (:337) and palindromic reads (:325) produce nothing.
                ok = i + mm <= ee && !pp;
            }
            append(ok && hash[w] < hashThreshold, hash[w], rr);               // :350, strict
        }
    }
    if(fill) flush();
}

// Unit seam: all window hashes of one kmer-id array.
__global__ void __launch_bounds__(256)
hashAllWindowsKernel(const uint32_t* __restrict__ kmerIds, uint64_t n, uint32_t m, uint64_t seed, uint64_t* __restrict__ out)
{
    const uint64_t j = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(j + m > n) return;
    uint32_t w[HASH_HALO + 1];
    for(uint32_t k = 0; k < m; k++) w[k] = kmerIds[j + k];
    out[j] = murmurWindow<0>(w, m, seed);
}

// ---------------------------------------------------------------------------
// K3/K4 helpers on the sorted records.
// ---------------------------------------------------------------------------
template<class K>
__global__ void __launch_bounds__(256)
markHeadsKernel(const K* __restrict__ keys, uint64_t n, uint32_t* __restrict__ flags)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
    else if(i == n) flags[i] = 0u;
}

// starts[g] = index of the first element of group g; starts[groupCount] = n.
template<class K>
__global__ void __launch_bounds__(256)
groupStartsKernel(const K* __restrict__ keys, const uint32_t* __restrict__ pos, uint64_t n, uint32_t* __restrict__ starts)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i < n) {
        if(i == 0 || keys[i] != keys[i - 1]) starts[pos[i]] = uint32_t(i);
    } else if(i == n) {
        starts[pos[n]] = uint32_t(n);
    }
}

constexpr int SIZE_HIST_CAP = 2048;

// Per record: pass-2 statistics (src/LowHash0.cpp:386-393), bucket-size histogram
// (:566-613, one vote per bucket = per head record), and the number of pairs this
// record starts in pass 3 (:430-457).
__global__ void __launch_bounds__(256)
bucketStatsKernel(
    const uint32_t* __restrict__ keys, const uint64_t* __restrict__ vals, const uint32_t* __restrict__ pos,
    const uint32_t* __restrict__ starts, uint64_t n,
    uint64_t minBucketSize, uint64_t maxBucketSize,
    unsigned long long* __restrict__ stats,             // [R][3]
    unsigned long long* __restrict__ sizeHist,          // [SIZE_HIST_CAP]
    uint32_t* __restrict__ overflowSizes, uint32_t* __restrict__ overflowCount, uint32_t overflowCapacity,
    uint64_t* __restrict__ pairCounts)                  // [n+1]
{
    __shared__ uint32_t sHist[SIZE_HIST_CAP];
    for(int k = threadIdx.x; k < SIZE_HIST_CAP; k += blockDim.x) sHist[k] = 0;
    __syncthreads();
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i < n) {
        const uint32_t b = pos[i + 1] - 1;
        const uint32_t begin = starts[b], end = starts[b + 1];
        const uint64_t size = end - begin;
        const uint64_t v = vals[i];
        const uint32_t orientedReadId = uint32_t(v);
        const uint32_t readId = orientedReadId >> 1;
        const int cls = (size < minBucketSize) ? 0 : ((size > maxBucketSize) ? 2 : 1);
        atomicAdd(&stats[3ULL * readId + cls], 1ULL);
        if(i == begin) {
            if(size < SIZE_HIST_CAP) atomicAdd(&sHist[size], 1u);
            else {
                const uint32_t o = atomicAdd(overflowCount, 1u);
                if(o < overflowCapacity) overflowSizes[o] = uint32_t(size);
            }
        }
        uint64_t count = 0;
        const uint64_t minSize = minBucketSize > 2 ? minBucketSize : 2;       // :436
        if(size >= minSize && size <= maxBucketSize) {
            const uint32_t hashHigh = uint32_t(v >> 32);
            for(uint32_t j = begin; j < end; j++) {
                const uint64_t u = vals[j];
                count += (uint32_t(u >> 32) == hashHigh && (uint32_t(u) >> 1) > readId) ? 1u : 0u;   // :443, :450
            }
        }
        pairCounts[i] = count;
    } else if(i == n) {
        pairCounts[i] = 0;
    }
    __syncthreads();
    for(int k = threadIdx.x; k < SIZE_HIST_CAP; k += blockDim.x) {
        const uint32_t c = sHist[k];
        if(c) atomicAdd(&sizeHist[k], (unsigned long long)c);
    }
}

// Pair key: readId0 | readId1 | strandBit packed so that integer order is the
// reference's (readId0, readId1, strand) order (src/LowHash0.hpp:131-134); strand
// bit 0 = same strand.
__global__ void __launch_bounds__(256)
pairWriteKernel(
    const uint64_t* __restrict__ vals, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ starts,
    const uint64_t* __restrict__ pairOffsets, uint64_t n, int readBits, uint64_t* __restrict__ pairKeys)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i >= n) return;
    uint64_t dst = pairOffsets[i];
    if(pairOffsets[i + 1] == dst) return;
    const uint32_t b = pos[i + 1] - 1;
    const uint32_t begin = starts[b], end = starts[b + 1];
    const uint64_t v = vals[i];
    const uint32_t hashHigh = uint32_t(v >> 32);
    const uint32_t o0 = uint32_t(v);
    const uint32_t readId0 = o0 >> 1;
    for(uint32_t j = begin; j < end; j++) {
        const uint64_t u = vals[j];
        const uint32_t o1 = uint32_t(u);
        if(uint32_t(u >> 32) == hashHigh && (o1 >> 1) > readId0) {
            pairKeys[dst++] = (uint64_t(readId0) << (readBits + 1)) | (uint64_t(o1 >> 1) << 1) | uint64_t((o0 ^ o1) & 1u);
        }
    }
}

// Run-length encode sorted pair keys into (key, count mod 2^16) appended at the
// end of the running table.  uint16 wrap: src/LowHash0.hpp:116, .cpp:521-555.
__global__ void __launch_bounds__(256)
runLengthKernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ starts, uint64_t groupCount,
    uint64_t* __restrict__ outKeys, uint32_t* __restrict__ outCounts)
{
    const uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(g >= groupCount) return;
    const uint32_t begin = starts[g], end = starts[g + 1];
    outKeys[g] = keys[begin];
    outCounts[g] = (end - begin) & 0xffffu;
}

// Fold groups of equal keys of the (sorted) table, summing frequencies mod 2^16.
__global__ void __launch_bounds__(256)
foldTableKernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts,
    const uint32_t* __restrict__ starts, uint64_t groupCount,
    uint64_t* __restrict__ outKeys, uint32_t* __restrict__ outCounts)
{
    const uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(g >= groupCount) return;
    const uint32_t begin = starts[g], end = starts[g + 1];
    uint32_t s = 0;
    for(uint32_t j = begin; j < end; j++) s += counts[j];
    outKeys[g] = keys[begin];
    outCounts[g] = s & 0xffffu;
}

__global__ void __launch_bounds__(256)
countHighFrequencyKernel(const uint32_t* __restrict__ counts, uint64_t n, uint32_t minFrequency, unsigned long long* __restrict__ out)
{
    uint32_t c = 0;
    for(uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x) {
        c += counts[i] >= minFrequency ? 1u : 0u;
    }
    for(int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, WAVE);
    if((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

__global__ void __launch_bounds__(256)
candidateFlagsKernel(const uint32_t* __restrict__ counts, uint64_t n, uint32_t minFrequency, uint32_t* __restrict__ flags)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i < n) flags[i] = counts[i] >= minFrequency ? 1u : 0u;
    else if(i == n) flags[i] = 0u;
}

// K6: src/LowHash0.cpp:204-214.
__global__ void __launch_bounds__(256)
emitCandidatesKernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pos, uint64_t n, int readBits,
    shasta_oriented_read_pair* __restrict__ out)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i >= n || pos[i + 1] == pos[i]) return;
    const uint64_t k = keys[i];
    shasta_oriented_read_pair p;
    p.readIds[0] = uint32_t(k >> (readBits + 1));
    p.readIds[1] = uint32_t((k >> 1) & ((1ULL << readBits) - 1ULL));
    p.isSameStrand = (k & 1ULL) ? 0 : 1;
    p.pad[0] = p.pad[1] = p.pad[2] = 0;
    out[pos[i]] = p;
}

// ---------------------------------------------------------------------------
// Context: markers in HBM.
// ---------------------------------------------------------------------------
Context::Context(int deviceArg) : device(deviceArg)
{
    int n = 0;
    HIP_CHECK(hipGetDeviceCount(&n));
    if(deviceArg < 0 || deviceArg >= n) throw std::runtime_error("shasta_mi355x: no such HIP device " + std::to_string(deviceArg));
    HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if(std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        throw std::runtime_error(std::string("shasta_mi355x is built for gfx950 only; device is ") + prop.gcnArchName);
    }
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
}

Context::~Context()
{
    (void)hipSetDevice(device);
    for(auto& scratch : alignScratch) scratch.reset();
    lowhashJob.reset();
    for(hipStream_t w : workerStream) if(w) (void)hipStreamDestroy(w);
    for(hipStream_t w : wideStream) if(w) (void)hipStreamDestroy(w);
    if(stream) (void)hipStreamDestroy(stream);
}

void Context::setMarkers(uint64_t readCountArg, const uint64_t* tocArg, const void* data7,
    const uint32_t* denseKmerIds, const uint8_t* flags, bool denseOnDevice)
{
    HIP_CHECK(hipSetDevice(device));
    MI355X_ASSERT(readCountArg < (1ULL << 31));
    readCount = readCountArg;
    const uint64_t orientedReadCount = 2 * readCount;
    hostToc.assign(tocArg, tocArg + orientedReadCount + 1);
    MI355X_ASSERT(hostToc[0] == 0);
    for(uint64_t i = 0; i < orientedReadCount; i++) MI355X_ASSERT(hostToc[i] <= hostToc[i + 1]);
    markerCount = hostToc[orientedReadCount];
    downsampled.reset();

    toc.reserve(orientedReadCount + 1, stream);
    HIP_CHECK(hipMemcpyAsync(toc.data(), hostToc.data(), (orientedReadCount + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    readFlags.reserve(std::max<uint64_t>(1, readCount), stream);
    if(flags) HIP_CHECK(hipMemcpyAsync(readFlags.data(), flags, readCount, hipMemcpyHostToDevice, stream));
    else HIP_CHECK(hipMemsetAsync(readFlags.data(), 0, std::max<uint64_t>(1, readCount), stream));

    kmerIds.reserve(markerCount + 2 * HASH_TILE + HASH_HALO, stream);      // the hash kernel reads whole tiles and window halos past the end
    if(markerCount) {
        if(denseKmerIds) {
            HIP_CHECK(hipMemcpyAsync(kmerIds.data(), denseKmerIds, markerCount * 4,
                denseOnDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
        } else {
            DeviceBuffer<uint32_t> packed;
            const uint64_t words = (7 * markerCount + 3) / 4 + 2;
            packed.reserve(words, stream);
            HIP_CHECK(hipMemsetAsync(packed.data() + (words - 3), 0, 3 * 4, stream));
            HIP_CHECK(hipMemcpyAsync(packed.data(), data7, 7 * markerCount, hipMemcpyHostToDevice, stream));
            const unsigned blocks = std::min<uint64_t>(divUp(markerCount, 256), 8192);
            hipLaunchKernelGGL(stripMarkersKernel, dim3(blocks), dim3(256), 0, stream,
                (const uint32_t*)packed.data(), kmerIds.data(), markerCount);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipStreamSynchronize(stream));
        }
    }
    const uint64_t tileCount = (markerCount + HASH_TILE - 1) / HASH_TILE;
    tileDesc.reserve(tileCount + 1, stream);
    hipLaunchKernelGGL(tileDescKernel, dim3(divUp(tileCount + 1, 256)), dim3(256), 0, stream,
        (const uint64_t*)toc.data(), (const uint8_t*)readFlags.data(), orientedReadCount, markerCount, tileDesc.data(), tileCount);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(stream));
}

// ---------------------------------------------------------------------------
// Host orchestration of the iteration loop.
// ---------------------------------------------------------------------------
namespace {

template<class T> T readDevice(const T* p, hipStream_t s)
{
    T v;
    HIP_CHECK(hipMemcpyAsync(&v, p, sizeof(T), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return v;
}

template<class T> T* mallocCopy(const std::vector<T>& v)
{
    T* p = static_cast<T*>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
    if(!p) throw std::bad_alloc();
    if(!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

int bitsFor(uint64_t maxValue) { int b = 1; while((maxValue >> b) != 0) ++b; return b; }

// The row of the kernel table a launch for this m is booked under (the template instance that runs).
const char* hashKernelName(uint32_t m)
{
    switch(m) {
        case 3: return "hashWindowsKernel<3>";
        case 4: return "hashWindowsKernel<4>";
        case 5: return "hashWindowsKernel<5>";
        default: return "hashWindowsKernel<0> (any m)";
    }
}

void launchHash(Context& ctx, uint32_t m, uint64_t seed, uint64_t threshold, uint32_t mask,
    uint64_t markerBegin, uint64_t markerEnd,
    uint32_t* outKeys, uint64_t* outVals, unsigned long long* counter, uint64_t capacity)
{
    const uint64_t tiles = (markerEnd + HASH_TILE - 1) / HASH_TILE - markerBegin / HASH_TILE;
    if(tiles == 0) return;
    // Persistent wavefronts: 256 CUs x 8 blocks x 4 independent wavefronts, each walking many tiles, so that one global
    // atomic serves a few hundred low hashes.
    const unsigned blocks = unsigned(std::min<uint64_t>(divUp(tiles, uint64_t(HASH_THREADS / WAVE)), 256 * 8));
#define SHASTA_LAUNCH_HASH(MF) hipLaunchKernelGGL(hashWindowsKernel<MF>, dim3(blocks), dim3(HASH_THREADS), 0, ctx.stream, \
        (const uint32_t*)ctx.kmerIds.data(), (const uint64_t*)ctx.toc.data(), (const uint8_t*)ctx.readFlags.data(), \
        (const uint4*)ctx.tileDesc.data(), markerBegin, markerEnd, ctx.markerCount, \
        m, seed, threshold, mask, outKeys, outVals, counter, capacity)
    switch(m) {
        case 3: SHASTA_LAUNCH_HASH(3); break;
        case 4: SHASTA_LAUNCH_HASH(4); break;
        case 5: SHASTA_LAUNCH_HASH(5); break;
        default: SHASTA_LAUNCH_HASH(0); break;
    }
#undef SHASTA_LAUNCH_HASH
    HIP_CHECK(hipGetLastError());
}

}  // namespace

// first index i in [0, n] with keys[i] >= bounds[k], for every k (keys sorted ascending).
template<class K>
__global__ void __launch_bounds__(64)
lowerBoundsKernel(const K* __restrict__ keys, uint64_t n, const K* __restrict__ bounds, uint32_t count, uint64_t* __restrict__ out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= count) return;
    const K bound = bounds[k];
    uint64_t lo = 0, hi = n;
    while(lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if(keys[mid] < bound) lo = mid + 1; else hi = mid;
    }
    out[k] = lo;
}

// ---------------------------------------------------------------------------
// The LowHash0 job: the state of one LowHash0::LowHash0 call, advanced in stages so that the
// same code runs on one GPU (lowhash0Run) and sharded over several (each rank hashes
// its own reads, owns a contiguous range of bucket ids and of readId0; the two exchange steps
// happen between the stages, in the caller).
//   hash(iteration)            K1 on the rank's reads, records sorted by bucket id, split by bucket owner
//   buckets(records)           K2-K5a on the records this rank owns: statistics, histogram, pair keys,
//                              sorted + run-length encoded, split by owner of readId0
//   merge(pairs)               K5b: fold the (key, count) runs this rank owns into its pair table
//   finish()                   K6: candidates of the rank's readId0 range, statistics (partial sums)
// ---------------------------------------------------------------------------
struct LowHash0Job {
    shasta_lowhash0_params p;
    int rank = 0, world = 1;
    std::vector<uint64_t> boundaries;       // world + 1 read ids: rank r owns readId0 in [boundaries[r], boundaries[r+1])
    uint64_t log2BucketCount = 0, bucketCount = 0, hashThreshold = 0;
    uint32_t mask = 0, minFrequency = 0;
    int readBits = 0, pairKeyBits = 0;
    uint64_t markerBegin = 0, markerEnd = 0;
    uint64_t recCapacity = 0, tableSize = 0;
    DeviceBuffer<uint32_t> recKeysA, recKeysB, flags, pos, starts, scanTemp32, overflowSizes, boundKeys32;
    DeviceBuffer<uint64_t> recValsA, recValsB, pairCounts, scanTemp64, pairKeysA, pairKeysB, tableKeysA, tableKeysB, runKeys, boundKeys64, boundOut;
    DeviceBuffer<uint32_t> tableCountsA, tableCountsB, runCounts;
    DeviceBuffer<unsigned long long> scalars, stats, sizeHist;
    DeviceBuffer<shasta_oriented_read_pair> candidatesDevice;
    static constexpr uint32_t overflowCapacity = 1 << 20;
};

namespace {
LowHash0Job& jobOf(Context& ctx)
{
    if(!ctx.lowhashJob) throw std::runtime_error("LowHash0: no job in progress (call begin first).");
    return *static_cast<LowHash0Job*>(ctx.lowhashJob.get());
}
}  // namespace

void lowhash0Begin(Context& ctx, const shasta_lowhash0_params& p, int rank, int world, const uint64_t* readBoundaries, uint32_t* log2BucketCountOut)
{
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;
    const uint64_t readCount = ctx.readCount;
    const uint64_t M = ctx.markerCount;
    if(p.m == 0 || p.m > HASH_HALO + 1) throw std::runtime_error("LowHash0: m must be in [1, 33].");
    if(readCount == 0) throw std::runtime_error("LowHash0: no reads.");
    if(world < 1 || rank < 0 || rank >= world) throw std::runtime_error("LowHash0: invalid rank / world size.");
    auto jobPtr = std::make_shared<LowHash0Job>();
    LowHash0Job& job = *jobPtr;
    job.p = p; job.rank = rank; job.world = world;
    job.boundaries.assign(size_t(world) + 1, 0);
    if(readBoundaries) job.boundaries.assign(readBoundaries, readBoundaries + world + 1);
    else { MI355X_ASSERT(world == 1); job.boundaries[0] = 0; job.boundaries[1] = readCount; }
    MI355X_ASSERT(job.boundaries[0] == 0 && job.boundaries[world] == readCount);
    for(int r = 0; r < world; r++) MI355X_ASSERT(job.boundaries[r] <= job.boundaries[r + 1]);

    // Bucket count, src/LowHash0.cpp:73-98 (from the marker count of ALL reads).
    const uint64_t estimate = referenceDoubleToUint64(p.hashFraction * double(M));
    const uint32_t log2Estimate = estimate ? 64 - uint32_t(__builtin_clzl(estimate)) : 0;
    uint64_t log2BucketCount = p.log2MinHashBucketCount;
    if(log2BucketCount == 0) log2BucketCount = 5 + log2Estimate;
    else if(log2BucketCount < log2Estimate) throw std::runtime_error("log2MinHashBucketCount is unreasonably small.");
    if(log2BucketCount > 31) log2BucketCount = 31;
    job.log2BucketCount = log2BucketCount;
    job.bucketCount = 1ULL << log2BucketCount;
    job.mask = uint32_t(job.bucketCount - 1);
    // :109
    job.hashThreshold = referenceDoubleToUint64(double(p.hashFraction) * double(std::numeric_limits<uint64_t>::max()));
    job.readBits = bitsFor(readCount - 1);
    job.pairKeyBits = 2 * job.readBits + 1;
    job.minFrequency = uint32_t(std::min<uint64_t>(p.minFrequency, 0x10000));   // frequency is uint16
    // This rank hashes the reads of its own range.
    job.markerBegin = ctx.hostToc[2 * job.boundaries[rank]];
    job.markerEnd = ctx.hostToc[2 * job.boundaries[rank + 1]];
    // First guess only (the hash stage grows it and repeats the iteration when it was too small); the fraction is
    // clamped because any double is a legal hashFraction in the reference (>= 1 keeps nothing, < 0 nearly everything).
    const double expectedFraction = !(p.hashFraction > 0.) ? 0. : std::min(p.hashFraction, 1.);
    job.recCapacity = std::max<uint64_t>(1 << 16, uint64_t(2.0 * expectedFraction * double(job.markerEnd - job.markerBegin)) + (1 << 16));

    job.scalars.reserve(8, stream);
    job.stats.reserve(3 * readCount, stream);
    job.sizeHist.reserve(SIZE_HIST_CAP, stream);
    job.overflowSizes.reserve(LowHash0Job::overflowCapacity + 1, stream);
    job.boundKeys32.reserve(size_t(world) + 1, stream); job.boundKeys64.reserve(size_t(world) + 1, stream);
    job.boundOut.reserve(size_t(world) + 1, stream);
    HIP_CHECK(hipMemsetAsync(job.stats.data(), 0, 3 * readCount * sizeof(unsigned long long), stream));
    // Split keys: first bucket id of every bucket owner, first pair key of every readId0 owner.
    std::vector<uint32_t> b32(size_t(world) + 1);
    std::vector<uint64_t> b64(size_t(world) + 1);
    for(int r = 0; r <= world; r++) {
        const uint64_t firstBucket = (uint64_t(r) * job.bucketCount + uint64_t(world) - 1) / uint64_t(world);
        b32[r] = uint32_t(std::min<uint64_t>(firstBucket, 0xffffffffULL));
        b64[r] = job.boundaries[r] << (job.readBits + 1);
    }
    HIP_CHECK(hipMemcpyAsync(job.boundKeys32.data(), b32.data(), b32.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(job.boundKeys64.data(), b64.data(), b64.size() * 8, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    ctx.lowhashJob = jobPtr;
    if(log2BucketCountOut) *log2BucketCountOut = uint32_t(log2BucketCount);
}

// Stage 1.  sendOffsets[r..r+1] delimit the records owned by rank r in (*keys, *vals).
void lowhash0Hash(Context& ctx, uint64_t iteration, uint64_t* sendOffsets, const uint32_t** keysOut, const uint64_t** valsOut)
{
    LowHash0Job& job = jobOf(ctx);
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;
    unsigned long long* counter = job.scalars.data();
    uint64_t n = 0;
    for(;;) {
        job.recKeysA.reserve(job.recCapacity, stream); job.recKeysB.reserve(job.recCapacity, stream);
        job.recValsA.reserve(job.recCapacity, stream); job.recValsB.reserve(job.recCapacity, stream);
        HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), stream));
        const KernelTimers::Span span = ctx.timers.begin(hashKernelName(uint32_t(job.p.m)), stream);
        launchHash(ctx, uint32_t(job.p.m), iteration * 37, job.hashThreshold, job.mask, job.markerBegin, job.markerEnd,
            job.recKeysA.data(), job.recValsA.data(), counter, job.recCapacity);
        const size_t handle = ctx.timers.end(span);
        n = readDevice(counter, stream);
        // Algorithmic bytes of the launch: 4 B per marker read + 12 B per low hash written; work = markers.
        ctx.timers.amend(handle, 4 * (job.markerEnd - job.markerBegin) + 12 * std::min(n, job.recCapacity), job.markerEnd - job.markerBegin);
        if(n <= job.recCapacity) break;
        job.recCapacity = n + n / 4;           // estimate was too small: grow and redo this iteration
    }
    MI355X_ASSERT(n < (1ULL << 32) - 1);
    // K2: bucket the records (radix partition on the bucket id); bucket owners are contiguous.
    const uint32_t* keys = job.recKeysA.data(); const uint64_t* vals = job.recValsA.data();
    {
        // K2: 12 bytes per record read + written per 8-bit pass.
        const uint64_t passes = (job.log2BucketCount + 7) / 8;
        const KernelTimers::Span span = ctx.timers.begin("radix sort of low-hash records", stream);
        if(radixSort<uint32_t, uint64_t, true>(job.recKeysA.data(), job.recKeysB.data(), job.recValsA.data(), job.recValsB.data(),
            n, int(job.log2BucketCount), ctx.sortWs, stream)) {
            keys = job.recKeysB.data(); vals = job.recValsB.data();
        }
        (void)ctx.timers.end(span, 2 * 12 * n * passes, n);
    }
    if(job.world == 1) {
        sendOffsets[0] = 0; sendOffsets[1] = n;
    } else {
        hipLaunchKernelGGL(lowerBoundsKernel<uint32_t>, dim3(divUp(uint64_t(job.world) + 1, 64)), dim3(64), 0, stream,
            keys, n, (const uint32_t*)job.boundKeys32.data(), uint32_t(job.world + 1), job.boundOut.data());
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(sendOffsets, job.boundOut.data(), (size_t(job.world) + 1) * 8, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        sendOffsets[0] = 0; sendOffsets[job.world] = n;     // the last bound may exceed 32 bits when log2 = 32 is approached
    }
    *keysOut = keys; *valsOut = vals;
}

// Stage 2.  (keys, vals): the n records of the buckets this rank owns (device pointers; any
// order when world > 1).  Produces the run-length encoded pair keys, split by owner of readId0.
void lowhash0Buckets(Context& ctx, const uint32_t* keysIn, const uint64_t* valsIn, uint64_t n,
    uint64_t* sendOffsets, const uint64_t** runKeysOut, const uint32_t** runCountsOut,
    uint64_t* bucketsUsedOut, uint64_t* sizeHistogramOut /*SIZE_HIST_CAP*/, std::vector<uint32_t>& overflowOut)
{
    LowHash0Job& job = jobOf(ctx);
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;
    const shasta_lowhash0_params& p = job.p;
    MI355X_ASSERT(n < (1ULL << 32) - 1);
    const uint32_t* keys = keysIn; const uint64_t* vals = valsIn;
    if(job.world > 1 && n) {
        // Concatenation of one sorted run per sender: sort again.
        job.recKeysA.reserve(n, stream); job.recKeysB.reserve(n, stream); job.recValsA.reserve(n, stream); job.recValsB.reserve(n, stream);
        if(keysIn != job.recKeysA.data()) HIP_CHECK(hipMemcpyAsync(job.recKeysA.data(), keysIn, n * 4, hipMemcpyDeviceToDevice, stream));
        if(valsIn != job.recValsA.data()) HIP_CHECK(hipMemcpyAsync(job.recValsA.data(), valsIn, n * 8, hipMemcpyDeviceToDevice, stream));
        keys = job.recKeysA.data(); vals = job.recValsA.data();
        if(radixSort<uint32_t, uint64_t, true>(job.recKeysA.data(), job.recKeysB.data(), job.recValsA.data(), job.recValsB.data(),
            n, int(job.log2BucketCount), ctx.sortWs, stream)) {
            keys = job.recKeysB.data(); vals = job.recValsB.data();
        }
    }

    // K3: bucket boundaries, statistics, histogram, pair counts.
    uint64_t bucketsUsed = 0, pairCount = 0;
    uint32_t* overflowCount = job.overflowSizes.data() + LowHash0Job::overflowCapacity;
    HIP_CHECK(hipMemsetAsync(job.sizeHist.data(), 0, SIZE_HIST_CAP * sizeof(unsigned long long), stream));
    HIP_CHECK(hipMemsetAsync(overflowCount, 0, 4, stream));
    if(n) {
        job.flags.reserve(n + 1, stream); job.pos.reserve(n + 1, stream); job.starts.reserve(n + 2, stream);
        job.scanTemp32.reserve(scanTempElements(n + 1), stream);
        job.pairCounts.reserve(n + 1, stream); job.scanTemp64.reserve(scanTempElements(n + 1), stream);
        const unsigned g = divUp(n + 1, 256);
        SHASTA_TIMED(ctx, "bucket boundaries (heads, scan, starts)", stream, 12 * n, n,
            hipLaunchKernelGGL(markHeadsKernel<uint32_t>, dim3(g), dim3(256), 0, stream, keys, n, job.flags.data());
            exclusiveScan<uint32_t>(job.flags.data(), job.pos.data(), n + 1, job.scanTemp32.data(), stream);
            hipLaunchKernelGGL(groupStartsKernel<uint32_t>, dim3(g), dim3(256), 0, stream,
                keys, (const uint32_t*)job.pos.data(), n, job.starts.data()));
        SHASTA_TIMED(ctx, "bucketStatsKernel + scan of pair counts", stream, 12 * n, n,
            hipLaunchKernelGGL(bucketStatsKernel, dim3(g), dim3(256), 0, stream,
                keys, vals, (const uint32_t*)job.pos.data(), (const uint32_t*)job.starts.data(), n,
                p.minBucketSize, p.maxBucketSize, job.stats.data(), job.sizeHist.data(),
                job.overflowSizes.data(), overflowCount, LowHash0Job::overflowCapacity, job.pairCounts.data());
            exclusiveScan<uint64_t>(job.pairCounts.data(), job.pairCounts.data(), n + 1, job.scanTemp64.data(), stream));
        HIP_CHECK(hipGetLastError());
        bucketsUsed = readDevice(job.pos.data() + n, stream);
        pairCount = readDevice(job.pairCounts.data() + n, stream);
    }
    *bucketsUsedOut = bucketsUsed;
    {
        static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "u64");
        HIP_CHECK(hipMemcpyAsync(sizeHistogramOut, job.sizeHist.data(), SIZE_HIST_CAP * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        const uint32_t overflow = readDevice(overflowCount, stream);
        if(overflow > LowHash0Job::overflowCapacity) throw std::runtime_error("LowHash0: bucket-size overflow list exhausted.");
        overflowOut.resize(overflow);
        if(overflow) {
            HIP_CHECK(hipMemcpyAsync(overflowOut.data(), job.overflowSizes.data(), overflow * 4ULL, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
        }
    }

    // K4 + K5a: pair keys, sorted, run-length encoded.
    uint64_t uniqueCount = 0;
    if(pairCount) {
        MI355X_ASSERT(pairCount < (1ULL << 32) - 1);
        job.pairKeysA.reserve(pairCount, stream); job.pairKeysB.reserve(pairCount, stream);
        SHASTA_TIMED(ctx, "pairWriteKernel", stream, 8 * pairCount, pairCount,
            hipLaunchKernelGGL(pairWriteKernel, dim3(divUp(n, 256)), dim3(256), 0, stream,
                vals, (const uint32_t*)job.pos.data(), (const uint32_t*)job.starts.data(),
                (const uint64_t*)job.pairCounts.data(), n, job.readBits, job.pairKeysA.data()));
        uint64_t* pk = job.pairKeysA.data();
        {
            const KernelTimers::Span span = ctx.timers.begin("radix sort of pair keys", stream);
            if(radixSort<uint64_t, uint32_t, false>(job.pairKeysA.data(), job.pairKeysB.data(), nullptr, nullptr, pairCount, job.pairKeyBits, ctx.sortWs, stream)) {
                pk = job.pairKeysB.data();
            }
            (void)ctx.timers.end(span, 2 * 8 * pairCount * uint64_t((job.pairKeyBits + 7) / 8), pairCount);
        }
        const KernelTimers::Span runSpan = ctx.timers.begin("run lengths of pair keys (heads, scan, starts, runs)", stream);
        job.flags.reserve(pairCount + 1, stream); job.pos.reserve(pairCount + 1, stream); job.starts.reserve(pairCount + 2, stream);
        job.scanTemp32.reserve(scanTempElements(pairCount + 1), stream);
        const unsigned g = divUp(pairCount + 1, 256);
        hipLaunchKernelGGL(markHeadsKernel<uint64_t>, dim3(g), dim3(256), 0, stream, (const uint64_t*)pk, pairCount, job.flags.data());
        exclusiveScan<uint32_t>(job.flags.data(), job.pos.data(), pairCount + 1, job.scanTemp32.data(), stream);
        hipLaunchKernelGGL(groupStartsKernel<uint64_t>, dim3(g), dim3(256), 0, stream,
            (const uint64_t*)pk, (const uint32_t*)job.pos.data(), pairCount, job.starts.data());
        HIP_CHECK(hipGetLastError());
        uniqueCount = readDevice(job.pos.data() + pairCount, stream);
        job.runKeys.reserve(uniqueCount, stream); job.runCounts.reserve(uniqueCount, stream);
        hipLaunchKernelGGL(runLengthKernel, dim3(divUp(uniqueCount, 256)), dim3(256), 0, stream,
            (const uint64_t*)pk, (const uint32_t*)job.starts.data(), uniqueCount, job.runKeys.data(), job.runCounts.data());
        HIP_CHECK(hipGetLastError());
        (void)ctx.timers.end(runSpan, 8 * pairCount + 12 * uniqueCount, pairCount);
    }
    if(job.world == 1 || uniqueCount == 0) {
        for(int r = 0; r <= job.world; r++) sendOffsets[r] = (r == job.world) ? uniqueCount : 0;
        if(job.world > 1) for(int r = 1; r < job.world; r++) sendOffsets[r] = 0;
    } else {
        hipLaunchKernelGGL(lowerBoundsKernel<uint64_t>, dim3(divUp(uint64_t(job.world) + 1, 64)), dim3(64), 0, stream,
            (const uint64_t*)job.runKeys.data(), uniqueCount, (const uint64_t*)job.boundKeys64.data(), uint32_t(job.world + 1), job.boundOut.data());
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(sendOffsets, job.boundOut.data(), (size_t(job.world) + 1) * 8, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        sendOffsets[0] = 0; sendOffsets[job.world] = uniqueCount;
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    *runKeysOut = job.runKeys.data(); *runCountsOut = job.runCounts.data();
}

// Stage 3.  (runKeys, runCounts): n (key, count) runs whose readId0 this rank owns.
void lowhash0Merge(Context& ctx, const uint64_t* runKeys, const uint32_t* runCounts, uint64_t n, uint64_t* highFrequencyOut, uint64_t* tableSizeOut)
{
    LowHash0Job& job = jobOf(ctx);
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;
    if(n) {
        const KernelTimers::Span mergeSpan = ctx.timers.begin("pair table merge (append + sort + fold)", stream);
        const uint64_t merged = job.tableSize + n;
        MI355X_ASSERT(merged < (1ULL << 32) - 1);
        job.tableKeysA.reserve(merged, stream, true); job.tableCountsA.reserve(merged, stream, true);
        job.tableKeysB.reserve(merged, stream); job.tableCountsB.reserve(merged, stream);
        HIP_CHECK(hipMemcpyAsync(job.tableKeysA.data() + job.tableSize, runKeys, n * 8, hipMemcpyDeviceToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(job.tableCountsA.data() + job.tableSize, runCounts, n * 4, hipMemcpyDeviceToDevice, stream));
        if(job.tableSize == 0 && job.world == 1) {
            job.tableSize = n;          // one sender: already sorted and unique
        } else {
            uint64_t* tk = job.tableKeysA.data(); uint32_t* tc = job.tableCountsA.data();
            uint64_t* ok = job.tableKeysB.data(); uint32_t* oc = job.tableCountsB.data();
            if(radixSort<uint64_t, uint32_t, true>(job.tableKeysA.data(), job.tableKeysB.data(), job.tableCountsA.data(), job.tableCountsB.data(),
                merged, job.pairKeyBits, ctx.sortWs, stream)) {
                std::swap(tk, ok); std::swap(tc, oc);
            }
            job.flags.reserve(merged + 1, stream); job.pos.reserve(merged + 1, stream); job.starts.reserve(merged + 2, stream);
            job.scanTemp32.reserve(scanTempElements(merged + 1), stream);
            const unsigned gm = divUp(merged + 1, 256);
            hipLaunchKernelGGL(markHeadsKernel<uint64_t>, dim3(gm), dim3(256), 0, stream, (const uint64_t*)tk, merged, job.flags.data());
            exclusiveScan<uint32_t>(job.flags.data(), job.pos.data(), merged + 1, job.scanTemp32.data(), stream);
            hipLaunchKernelGGL(groupStartsKernel<uint64_t>, dim3(gm), dim3(256), 0, stream,
                (const uint64_t*)tk, (const uint32_t*)job.pos.data(), merged, job.starts.data());
            HIP_CHECK(hipGetLastError());
            const uint64_t folded = readDevice(job.pos.data() + merged, stream);
            hipLaunchKernelGGL(foldTableKernel, dim3(divUp(folded, 256)), dim3(256), 0, stream,
                (const uint64_t*)tk, (const uint32_t*)tc, (const uint32_t*)job.starts.data(), folded, ok, oc);
            HIP_CHECK(hipGetLastError());
            // Result is in (ok, oc); make it the A side.
            if(ok != job.tableKeysA.data()) { job.tableKeysA.swap(job.tableKeysB); job.tableCountsA.swap(job.tableCountsB); }
            job.tableSize = folded;
        }
        (void)ctx.timers.end(mergeSpan, 12 * merged, merged);
    }
    // Per-iteration summary (src/LowHash0.cpp:184-196): this rank's share.
    uint64_t highFrequency = 0;
    if(job.tableSize) {
        unsigned long long* highCounter = job.scalars.data() + 1;
        HIP_CHECK(hipMemsetAsync(highCounter, 0, sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(countHighFrequencyKernel, dim3(std::min<unsigned>(divUp(job.tableSize, 256), 2048)), dim3(256), 0, stream,
            (const uint32_t*)job.tableCountsA.data(), job.tableSize, job.minFrequency, highCounter);
        HIP_CHECK(hipGetLastError());
        highFrequency = readDevice(highCounter, stream);
    }
    *highFrequencyOut = highFrequency; *tableSizeOut = job.tableSize;
}

// Stage 4.  Candidates of this rank's readId0 range (sorted), statistics (this rank's partial
// sums, readCount x 3), hash-kernel timing; ends the job.
void lowhash0Finish(Context& ctx, uint64_t* readLowHashStatistics, std::vector<shasta_oriented_read_pair>& hostCandidates)
{
    LowHash0Job& job = jobOf(ctx);
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;
    const uint64_t readCount = ctx.readCount;
    hostCandidates.clear();
    // K6.
    if(job.tableSize) {
        job.flags.reserve(job.tableSize + 1, stream); job.pos.reserve(job.tableSize + 1, stream);
        job.scanTemp32.reserve(scanTempElements(job.tableSize + 1), stream);
        const unsigned g = divUp(job.tableSize + 1, 256);
        hipLaunchKernelGGL(candidateFlagsKernel, dim3(g), dim3(256), 0, stream,
            (const uint32_t*)job.tableCountsA.data(), job.tableSize, job.minFrequency, job.flags.data());
        exclusiveScan<uint32_t>(job.flags.data(), job.pos.data(), job.tableSize + 1, job.scanTemp32.data(), stream);
        HIP_CHECK(hipGetLastError());
        const uint64_t candidateCount = readDevice(job.pos.data() + job.tableSize, stream);
        if(candidateCount) {
            job.candidatesDevice.reserve(candidateCount, stream);
            hipLaunchKernelGGL(emitCandidatesKernel, dim3(g), dim3(256), 0, stream,
                (const uint64_t*)job.tableKeysA.data(), (const uint32_t*)job.pos.data(), job.tableSize, job.readBits, job.candidatesDevice.data());
            HIP_CHECK(hipGetLastError());
            hostCandidates.resize(candidateCount);
            HIP_CHECK(hipMemcpyAsync(hostCandidates.data(), job.candidatesDevice.data(),
                candidateCount * sizeof(shasta_oriented_read_pair), hipMemcpyDeviceToHost, stream));
        }
    }
    HIP_CHECK(hipMemcpyAsync(readLowHashStatistics, job.stats.data(), 3 * readCount * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));

    ctx.lowhashJob.reset();
}

// Histogram rows (src/LowHash0.cpp:586-595) of one iteration from the (summed) size histogram.
static void appendHistogramRows(uint64_t iteration, uint64_t bucketCount, uint64_t bucketsUsed,
    const uint64_t* sizeHistogram, const std::vector<uint32_t>& overflow, std::vector<uint64_t>& histogramRows)
{
    std::map<uint64_t, uint64_t> rows;
    if(bucketCount > bucketsUsed) rows[0] = bucketCount - bucketsUsed;
    for(int s = 1; s < SIZE_HIST_CAP; s++) if(sizeHistogram[s]) rows[uint64_t(s)] = sizeHistogram[s];
    for(uint32_t s : overflow) ++rows[s];
    for(const auto& r : rows) { histogramRows.push_back(iteration); histogramRows.push_back(r.first); histogramRows.push_back(r.second); }
}

// The whole of LowHash0::LowHash0 on one GPU: the stages above, no exchange in between.
void lowhash0Run(Context& ctx, const shasta_lowhash0_params& p, uint64_t* readLowHashStatistics, shasta_lowhash0_result& result)
{
    std::memset(&result, 0, sizeof(result));
    const auto t0 = std::chrono::steady_clock::now();
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;
    const uint64_t readCount = ctx.readCount;
    hipEvent_t evBegin, evEnd;
    HIP_CHECK(hipEventCreate(&evBegin)); HIP_CHECK(hipEventCreate(&evEnd));
    HIP_CHECK(hipEventRecord(evBegin, stream));
    uint32_t log2BucketCount = 0;
    lowhash0Begin(ctx, p, 0, 1, nullptr, &log2BucketCount);
    result.log2BucketCount = log2BucketCount;
    const uint64_t bucketCount = 1ULL << log2BucketCount;

    std::vector<uint64_t> highFrequencyPerIteration, totalPerIteration, histogramRows, sizeHistogram(SIZE_HIST_CAP);
    std::vector<uint32_t> overflow;
    std::vector<shasta_oriented_read_pair> hostCandidates;
    try {
        uint64_t highFrequency = 0;
        for(uint64_t iteration = 0; ; iteration++) {
            // Iteration control, src/LowHash0.cpp:136-157.
            if(p.minHashIterationCount == 0) {
                const double current = 2. * double(highFrequency) / double(readCount);
                if(current >= p.alignmentCandidatesPerRead) break;
            } else if(iteration == p.minHashIterationCount) {
                break;
            }
            uint64_t offsets[2];
            const uint32_t* keys = nullptr; const uint64_t* vals = nullptr;
            lowhash0Hash(ctx, iteration, offsets, &keys, &vals);
            const uint64_t* runKeys = nullptr; const uint32_t* runCounts = nullptr;
            uint64_t bucketsUsed = 0;
            lowhash0Buckets(ctx, keys, vals, offsets[1], offsets, &runKeys, &runCounts, &bucketsUsed, sizeHistogram.data(), overflow);
            appendHistogramRows(iteration, bucketCount, bucketsUsed, sizeHistogram.data(), overflow, histogramRows);
            uint64_t tableSize = 0;
            lowhash0Merge(ctx, runKeys, runCounts, offsets[1], &highFrequency, &tableSize);
            highFrequencyPerIteration.push_back(highFrequency);
            totalPerIteration.push_back(tableSize);
        }
        lowhash0Finish(ctx, readLowHashStatistics, hostCandidates);
    } catch(...) {
        ctx.lowhashJob.reset();
        (void)hipEventDestroy(evBegin); (void)hipEventDestroy(evEnd);
        throw;
    }
    HIP_CHECK(hipEventRecord(evEnd, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, evBegin, evEnd));
    result.deviceSeconds = ms * 1e-3;
    (void)hipEventDestroy(evBegin); (void)hipEventDestroy(evEnd);

    result.candidateCount = hostCandidates.size();
    result.candidates = mallocCopy(hostCandidates);
    result.iterationCount = uint32_t(highFrequencyPerIteration.size());
    result.highFrequency = mallocCopy(highFrequencyPerIteration);
    result.total = mallocCopy(totalPerIteration);
    result.histogramRowCount = histogramRows.size() / 3;
    result.histogram = mallocCopy(histogramRows);
    result.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void lowhash0Free(shasta_lowhash0_result& r)
{
    std::free(r.candidates); std::free(r.highFrequency); std::free(r.total); std::free(r.histogram);
    std::memset(&r, 0, sizeof(r));
}

// ---------------------------------------------------------------------------
// PMC calibration (MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE must be calibrated
// on a known byte count in the kernel's own access pattern).  Reads `bytes` with the hash
// kernel's pattern (one dword per lane, 256-marker tiles) or writes `bytes` with the DP trace's
// pattern (four lanes of a wave store 8 bytes each per record).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
calibrateReadDwordKernel(const uint32_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ out)
{
    uint32_t acc = 0;
    for(uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) acc ^= in[i];
    if(acc == 0x12345678u) out[0] = acc;      // keeps the loads alive; practically never true
}

__global__ void __launch_bounds__(256)
calibrateWriteRecordKernel(uint64_t* __restrict__ out, uint64_t records)
{
    const uint64_t wave = (uint64_t(blockIdx.x) * 256 + threadIdx.x) >> 6, waves = (uint64_t(gridDim.x) * 256) >> 6;
    const int lane = int(threadIdx.x) & 63;
    for(uint64_t r = wave; r < records; r += waves) if(lane < 4) out[4 * r + lane] = r + lane;
}

void calibrateUnit(uint64_t bytes, int mode)
{
    int count = 0;
    HIP_CHECK(hipGetDeviceCount(&count));
    if(count == 0) throw std::runtime_error("shasta_mi355x: no HIP device.");
    DeviceBuffer<uint32_t> buffer;
    buffer.reserve(bytes / 4 + 64);
    HIP_CHECK(hipMemset(buffer.data(), 1, bytes));
    HIP_CHECK(hipDeviceSynchronize());
    if(mode == 0) {
        hipLaunchKernelGGL(calibrateReadDwordKernel, dim3(2048), dim3(256), 0, nullptr, (const uint32_t*)buffer.data(), bytes / 4, buffer.data() + bytes / 4);
    } else {
        hipLaunchKernelGGL(calibrateWriteRecordKernel, dim3(2048), dim3(256), 0, nullptr, reinterpret_cast<uint64_t*>(buffer.data()), bytes / 32);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
}

void hashWindowsUnit(const uint32_t* kmerIds, uint64_t n, uint64_t m, uint64_t iteration, uint64_t* out)
{
    if(m == 0 || m > HASH_HALO + 1) throw std::runtime_error("hash_windows: m must be in [1, 33].");
    if(n < m) return;
    int count = 0;
    HIP_CHECK(hipGetDeviceCount(&count));
    if(count == 0) throw std::runtime_error("shasta_mi355x: no HIP device.");
    DeviceBuffer<uint32_t> in; DeviceBuffer<uint64_t> o;
    in.reserve(n); o.reserve(n);
    HIP_CHECK(hipMemcpy(in.data(), kmerIds, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(hashAllWindowsKernel, dim3(divUp(n, 256)), dim3(256), 0, nullptr,
        (const uint32_t*)in.data(), n, uint32_t(m), iteration * 37, o.data());
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpy(out, o.data(), (n - m + 1) * 8, hipMemcpyDeviceToHost));
}

}  // namespace shasta_mi355x
