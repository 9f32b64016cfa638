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

// Every stride-th marker, for Context::setMarkers' estimate of how often two markers of the read set are equal.
__global__ void __launch_bounds__(256)
sampleMarkersKernel(const uint32_t* __restrict__ kmerIds, uint64_t stride, uint32_t count, uint32_t* __restrict__ sample)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < count) sample[i] = kmerIds[uint64_t(i) * stride];
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

// ALL (round 3): every MinHash iteration of the job in ONE pass over the markers.  The seed only enters a window's hash through
// h0 = seed ^ (len mul) (src/MurmurHash2.cpp:96-140): the block transforms murmurMix(block) -- two of the five 64 x 64-bit
// multiplies per window and iteration at m = 4 -- the read boundaries, the palindromic flags and the marker loads do not depend
// on it.  So a tile is loaded and prepared once and hashed `iterations` times (seed = 37 t, :316): 8 + 12 iterations multiplies
// per lane and tile instead of 20 iterations, 4 M bytes read instead of 4 M iterations, one launch instead of `iterations`.
// The records of iteration t carry t above the bucket id -- key = t << iterationShift | bucket id -- so that ONE sort leaves
// them grouped by (iteration, bucket) and the bucket kernels run once over all of them.  `seed` is unused then.
// KEY: the records' key type -- uint32_t wherever iteration | bucket id fits 32 bits; uint64_t (iteration << 32 | bucket id)
// beyond, e.g. at 2^31 buckets, the human-genome value.
template<int M_FIXED, bool ALL = false, class KEY = uint32_t>
__global__ void __launch_bounds__(HASH_THREADS)
hashWindowsKernel(
    const uint32_t* __restrict__ kmerIds, const uint64_t* __restrict__ toc,
    const uint8_t* __restrict__ readFlags, const uint4* __restrict__ tileDesc,
    uint64_t markerBegin, uint64_t markerEnd, uint64_t markerCount,
    uint32_t m, uint64_t seed, uint64_t hashThreshold, uint32_t mask,
    KEY* __restrict__ outKeys, uint64_t* __restrict__ outVals,
    unsigned long long* __restrict__ counter, uint64_t capacity, uint32_t iterations, uint32_t iterationShift)
{
    __shared__ KEY stageKeys[HASH_THREADS / WAVE][HASH_STAGE];
    __shared__ uint64_t stageVals[HASH_THREADS / WAVE][HASH_STAGE];
    constexpr uint64_t mul = 0xc6a4a7935bd1e995ULL;
    const uint32_t mm = M_FIXED ? uint32_t(M_FIXED) : m;
    const int lane = laneId();
    const uint32_t waveInBlock = threadIdx.x >> 6;
    KEY* const sKeys = stageKeys[waveInBlock];
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
    auto append = [&](bool hit, uint64_t hash, uint32_t orientedReadId, KEY iterationBits) {
        const uint64_t votes = __ballot(hit);
        if(votes == 0) return;
        if constexpr (ALL) { if(fill + uint32_t(WAVE) > uint32_t(HASH_STAGE)) flush(); }       // (a tile adds up to HASH_TILE records per iteration)
        if(hit) {
            const uint32_t slot = fill + uint32_t(__popcll(votes & laneMaskLt()));
            sKeys[slot] = KEY(uint32_t(hash) & mask) | iterationBits;           // bucket id, :352 (under the iteration, ALL)
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
        if(!ALL && fill + HASH_TILE > HASH_STAGE) flush();                    // room for every window of this tile
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

        // What does not depend on the seed: the lane's markers and its neighbour's, the block transforms, and (below) which
        // of the four windows count.
        uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0, k6 = 0, k7 = 0;
        uint64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0;
        if constexpr (M_FIXED >= 3 && M_FIXED <= 5) {
            // Markers 0..3 are the lane's own, 4..7 the next lane's.
            k0 = K.x; k1 = K.y; k2 = K.z; k3 = K.w;
            k4 = fromNextLane(K.x); k5 = fromNextLane(K.y); k6 = fromNextLane(K.z); k7 = fromNextLane(K.w);
            // Block transforms that start at the lane's own markers, and the next lane's first two.
            p0 = murmurMix(uint64_t(k0) | (uint64_t(k1) << 32)); p1 = murmurMix(uint64_t(k1) | (uint64_t(k2) << 32));
            p2 = murmurMix(uint64_t(k2) | (uint64_t(k3) << 32)); p3 = murmurMix(uint64_t(k3) | (uint64_t(k4) << 32));
            if constexpr (M_FIXED >= 4) { p4 = fromNextLane64(p0); p5 = fromNextLane64(p1); }
        }
        (void)k0; (void)k1; (void)k5; (void)k6; (void)k7; (void)p4; (void)p5;
        auto hashes = [&](uint64_t seedNow, uint64_t (&hash)[4]) {
        if constexpr (M_FIXED >= 3 && M_FIXED <= 5) {
            const uint64_t h0 = seedNow ^ (uint64_t(4u * M_FIXED) * mul);
            auto finish = [&](uint64_t h) { h ^= h >> 47; h *= mul; h ^= h >> 47; return h; };
            if constexpr (M_FIXED == 3) {
                hash[0] = finish(((h0 ^ p0) * mul ^ uint64_t(k2)) * mul);
                hash[1] = finish(((h0 ^ p1) * mul ^ uint64_t(k3)) * mul);
                hash[2] = finish(((h0 ^ p2) * mul ^ uint64_t(k4)) * mul);
                hash[3] = finish(((h0 ^ p3) * mul ^ uint64_t(k5)) * mul);
            } else if constexpr (M_FIXED == 4) {
                hash[0] = finish(((h0 ^ p0) * mul ^ p2) * mul);
                hash[1] = finish(((h0 ^ p1) * mul ^ p3) * mul);
                hash[2] = finish(((h0 ^ p2) * mul ^ p4) * mul);
                hash[3] = finish(((h0 ^ p3) * mul ^ p5) * mul);
            } else {
                hash[0] = finish((((h0 ^ p0) * mul ^ p2) * mul ^ uint64_t(k4)) * mul);
                hash[1] = finish((((h0 ^ p1) * mul ^ p3) * mul ^ uint64_t(k5)) * mul);
                hash[2] = finish((((h0 ^ p2) * mul ^ p4) * mul ^ uint64_t(k6)) * mul);
                hash[3] = finish((((h0 ^ p3) * mul ^ p5) * mul ^ uint64_t(k7)) * mul);
            }
        } else {
            // Any m: the window's markers one by one (they are in L1 after the tile load).
#pragma unroll
            for(int w = 0; w < 4; w++) {
                uint32_t win[HASH_HALO + 1];
                for(uint32_t q = 0; q < mm; q++) win[q] = kmerIds[i0 + uint32_t(w) + q];
                hash[w] = murmurWindow<0>(win, mm, seedNow);
            }
        }
        };
        bool counts[4];
        uint32_t readOfWindow[4];
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
            counts[w] = ok; readOfWindow[w] = rr;
        }
        if constexpr (ALL) {
            for(uint32_t t = 0; t < iterations; t++) {
                uint64_t hash[4];
                hashes(uint64_t(t) * 37ULL, hash);                                // :316
#pragma unroll
                for(int w = 0; w < 4; w++) append(counts[w] && hash[w] < hashThreshold, hash[w], readOfWindow[w], KEY(t) << iterationShift);
            }
        } else {
            uint64_t hash[4];
            hashes(seed, hash);
#pragma unroll
            for(int w = 0; w < 4; w++) append(counts[w] && hash[w] < hashThreshold, hash[w], readOfWindow[w], KEY(0));             // :350, strict
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
// K3 + K4 on the records of one iteration sorted by bucket id: bucket boundaries (first-record flags, their scan, the
// start of every bucket), then per record the pass-2 statistics, one vote per bucket for the size histogram and the number
// of pairs the record starts, a scan of those, and the pair keys.  No host value is needed anywhere in between: counts
// stay on the device (primitives.hpp, Count), grids are sized from the capacity of the record buffers.
// (A single kernel that finds a record's bucket by walking its neighbours was tried: 1.09 ms per iteration against
// 0.35 ms for this chain -- dependent loads in the walk, and one atomic per wavefront on the append cursor.)
// ---------------------------------------------------------------------------
constexpr int SIZE_HIST_CAP = 2048;
constexpr int SIZE_HIST_LDS = 256;

// Device-side counters of a LowHash0 job.
enum : int {
    C_RECORDS = 0,          // low hashes kept by the hash kernel of the current iteration
    C_PAIRS = 1,            // pair keys appended so far (all iterations; or this iteration's, staged API)
    C_OVERFLOW = 2,         // entries of the list of bucket sizes beyond the histogram bins
    C_MAX_RECORDS = 3,      // largest C_RECORDS of any iteration
    C_BUCKETS = 4,          // buckets used by the current iteration
    C_TOTAL_RECORDS = 5,    // (all iterations in one pass) low hashes the hash kernel found in all
    C_COUNT = 8
};

// The sorted records' keys as the bucket kernels see them: 32-bit words, `words` per key (1: iteration << shift | bucket id;
// 2: a 64-bit key, bucket id in the low word, iteration in the high one, shift = 0).
// (A sharded job's 64-bit keys carry their owner in the top byte: owner << 56 | iteration << 32 | bucket id.)
struct RecordKeys {
    const uint32_t* words; uint32_t perKey, shift;
    __host__ __device__ uint32_t iteration(uint64_t i) const { return perKey == 2u ? (words[2u * i + 1u] & 0x00ffffffu) : words[i] >> shift; }
    __host__ __device__ bool differ(uint64_t i, uint64_t j) const { return words[i * perKey] != words[j * perKey] || (perKey == 2u && words[i * 2u + 1u] != words[j * 2u + 1u]); }
};

// flags[i] = 1 where a bucket begins; flags[n] = 0 (n = the exact count; entries past it are never looked at).
__global__ void __launch_bounds__(256)
markHeadsKernel(RecordKeys keys, Count count, uint32_t* __restrict__ flags)
{
    const uint64_t n = count.get();
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i < n) flags[i] = (i == 0 || keys.differ(i, i - 1)) ? 1u : 0u;
    else if(i == n) flags[i] = 0u;
}

// pos = exclusive scan of flags.  starts[g] = index of the first record of bucket g; starts[bucket count] = n.
__global__ void __launch_bounds__(256)
groupStartsKernel(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ pos, Count count, uint32_t* __restrict__ starts)
{
    const uint64_t n = count.get();
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i < n) {
        if(flags[i]) starts[pos[i]] = uint32_t(i);
    } else if(i == n) {
        starts[pos[n]] = uint32_t(n);
    }
}

// Per record: pass-2 statistics (src/LowHash0.cpp:386-393), bucket-size histogram (:566-613, one vote per bucket =
// per first record), and the number of pairs this record starts in pass 3 (:430-457).
// iterationKeys (records of all iterations in one array, key = iteration << iterationShift | bucket id): the iteration of a
// record is read off its key and sizeHist is the table of all rows; otherwise `iteration` and its row.
__global__ void __launch_bounds__(256)
bucketStatsKernel(
    const uint64_t* __restrict__ vals, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ starts, Count count,
    uint64_t minBucketSize, uint64_t maxBucketSize, uint32_t iteration, RecordKeys iterationKeys,
    unsigned long long* __restrict__ stats,             // [R][3]
    unsigned long long* __restrict__ sizeHist,          // [SIZE_HIST_CAP] of this iteration (of iteration 0 with iterationKeys)
    unsigned long long* __restrict__ overflowSizes, uint32_t overflowCapacity,     // iteration << 32 | size
    unsigned long long* __restrict__ counters,
    uint64_t* __restrict__ pairCounts,                  // [n+1]
    uint32_t* __restrict__ statKeys)                    // [n]: 3 readId + class of every record, for readStatisticsKernel; null: counted here, an atomic per record
{
    __shared__ uint32_t sHist[SIZE_HIST_LDS];
    const uint64_t n = count.get();
    sHist[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    // The workgroup's LDS histogram belongs to the iteration of its first record; a record of another one (a workgroup on an
    // iteration boundary) votes in its own row directly.
    const uint64_t blockFirst = uint64_t(blockIdx.x) * blockDim.x;
    const uint32_t blockIteration = (iterationKeys.words && blockFirst < n) ? iterationKeys.iteration(blockFirst) : iteration;
    if(iterationKeys.words) sizeHist += uint64_t(blockIteration) * SIZE_HIST_CAP;
    if(i < n) {
        if(iterationKeys.words) iteration = iterationKeys.iteration(i);
        unsigned long long* const myHist = sizeHist + (int64_t(iteration) - int64_t(blockIteration)) * SIZE_HIST_CAP;
        const uint32_t b = pos[i + 1] - 1;
        const uint32_t begin = starts[b], end = starts[b + 1];
        const uint64_t size = end - begin;
        const uint64_t v = vals[i];
        const uint32_t orientedReadId = uint32_t(v);
        const uint32_t readId = orientedReadId >> 1;
        const int cls = (size < minBucketSize) ? 0 : ((size > maxBucketSize) ? 2 : 1);
        if(statKeys) statKeys[i] = 3u * readId + uint32_t(cls);
        else atomicAdd(&stats[3ULL * readId + cls], 1ULL);
        if(i == begin) {
            if(size < SIZE_HIST_LDS && iteration == blockIteration) atomicAdd(&sHist[size], 1u);
            else if(size < SIZE_HIST_CAP) atomicAdd(&myHist[size], 1ULL);
            else {
                const unsigned long long o = atomicAdd(&counters[C_OVERFLOW], 1ULL);
                if(o < overflowCapacity) overflowSizes[o] = ((unsigned long long)iteration << 32) | (unsigned long long)(size < 0xffffffffULL ? size : 0xffffffffULL);
            }
        }
        uint64_t pairs = 0;
        const uint64_t minSize = minBucketSize > 2 ? minBucketSize : 2;       // :436
        if(size >= minSize && size <= maxBucketSize) {
            const uint32_t hashHigh = uint32_t(v >> 32);
            for(uint32_t j = begin; j < end; j++) {
                const uint64_t u = vals[j];
                pairs += (uint32_t(u >> 32) == hashHigh && (uint32_t(u) >> 1) > readId) ? 1u : 0u;   // :443, :450
            }
        }
        pairCounts[i] = pairs;
    } else if(i == n) {
        pairCounts[i] = 0;
    }
    __syncthreads();
    const uint32_t c = sHist[threadIdx.x];
    if(c) atomicAdd(&sizeHist[threadIdx.x], (unsigned long long)c);
}

// The pass-2 statistics (src/LowHash0.cpp:386-393: per read, its records by the class of their bucket's size) from the keys
// bucketStatsKernel wrote, 3 readId + class.  The records are in bucket order, so a read's records are anywhere: an atomic per
// record on the [R][3] table is 3e7 read-modify-writes of random lines at the memory side (the table is shared by the XCDs, so
// their L2s cannot hold it: 2.2 of the kernel's 2.4 ms, five times its other traffic).  Instead the keys are put in the order of
// their bits above STAT_LOW_BITS by the radix sort's passes (one pass at 100 k reads: 19-bit keys), after which a workgroup's
// span of keys lies in one partition of 2^STAT_LOW_BITS table entries, or a few: it counts them in LDS and adds what is not zero
// to the table when the partition changes and at its end -- 2^STAT_LOW_BITS atomics per span of STAT_SPAN keys at most.
constexpr int STAT_LOW_BITS = 11;
constexpr uint32_t STAT_SPAN = 1u << 16;
__global__ void __launch_bounds__(256)
readStatisticsKernel(const uint32_t* __restrict__ statKeys, Count count, unsigned long long* __restrict__ stats, uint64_t entries)
{
    __shared__ uint32_t counts[1 << STAT_LOW_BITS];
    __shared__ uint32_t nextPartition;
    const uint64_t n = count.get();
    const uint64_t first = uint64_t(blockIdx.x) * STAT_SPAN;
    if(first >= n) return;
    const uint64_t end = first + STAT_SPAN < n ? first + STAT_SPAN : n;
    for(uint32_t k = threadIdx.x; k < (1u << STAT_LOW_BITS); k += blockDim.x) counts[k] = 0;
    uint32_t partition = statKeys[first] >> STAT_LOW_BITS;          // (the same in every thread)
    __syncthreads();
    auto flush = [&]() {
        for(uint32_t k = threadIdx.x; k < (1u << STAT_LOW_BITS); k += blockDim.x) {
            const uint32_t c = counts[k];
            counts[k] = 0;
            const uint64_t entry = (uint64_t(partition) << STAT_LOW_BITS) + k;
            if(c && entry < entries) atomicAdd(&stats[entry], (unsigned long long)c);
        }
    };
    constexpr int PER_THREAD = 4;
    for(uint64_t base = first; base < end; base += uint64_t(PER_THREAD) * blockDim.x) {
        uint32_t key[PER_THREAD];
        bool pending[PER_THREAD];
#pragma unroll
        for(int u = 0; u < PER_THREAD; u++) {
            const uint64_t i = base + uint64_t(u) * blockDim.x + threadIdx.x;
            pending[u] = i < end;
            key[u] = pending[u] ? statKeys[i] : 0u;
        }
        // The keys ascend in their partition: the step's keys are counted partition by partition (one round nearly always).
        for(;;) {
            uint32_t least = 0xffffffffu;
#pragma unroll
            for(int u = 0; u < PER_THREAD; u++) {
                if(pending[u] && (key[u] >> STAT_LOW_BITS) == partition) { atomicAdd(&counts[key[u] & ((1u << STAT_LOW_BITS) - 1u)], 1u); pending[u] = false; }
                if(pending[u]) least = min(least, key[u] >> STAT_LOW_BITS);
            }
            if(threadIdx.x == 0) nextPartition = 0xffffffffu;
            __syncthreads();
            if(least != 0xffffffffu) atomicMin(&nextPartition, least);
            __syncthreads();
            const uint32_t next = nextPartition;                   // (every thread reads the same word: the loop is uniform)
            __syncthreads();
            if(next == 0xffffffffu) break;
            flush();
            partition = next;
            __syncthreads();
        }
    }
    __syncthreads();
    flush();
}

// Pass 3 (:424-459): every pair of records of an admissible bucket with equal hashHighBits and readId0 < readId1, as a
// 64-bit key readId0 | readId1 | strandBit whose integer order is the reference's (readId0, readId1, strand) order
// (src/LowHash0.hpp:131-134; strand bit 0 = same strand), appended after the keys of the earlier iterations
// (counters[C_PAIRS]) and tagged with the iteration.  pairOffsets = exclusive scan of the counts above.
__global__ void __launch_bounds__(256)
pairWriteKernel(
    const uint64_t* __restrict__ vals, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ starts,
    const uint64_t* __restrict__ pairOffsets, Count count, int readBits, uint32_t iteration, RecordKeys iterationKeys,
    const unsigned long long* __restrict__ counters, uint64_t* __restrict__ pairKeys, uint32_t* __restrict__ pairTags, uint64_t pairCapacity)
{
    const uint64_t n = count.get();
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i >= n) return;
    uint64_t dst = pairOffsets[i];
    if(pairOffsets[i + 1] == dst) return;
    if(iterationKeys.words) iteration = iterationKeys.iteration(i);
    dst += counters[C_PAIRS];
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
            if(dst < pairCapacity) {                    // past the capacity the keys are dropped, the cursor stays exact: the job is run again with room
                pairKeys[dst] = (uint64_t(readId0) << (readBits + 1)) | (uint64_t(o1 >> 1) << 1) | uint64_t((o0 ^ o1) & 1u);
                if(pairTags) pairTags[dst] = iteration;
            }
            ++dst;
        }
    }
}

// After the kernels of an iteration: its counters into the per-iteration table (rows of 4: records, buckets used,
// pair cursor at its end, unused), the append cursor moved past its keys, the running maximum of the record count;
// the record counter starts again.  pos / pairOffsets: the two scans above (null when there was nothing to scan).
__global__ void noteIterationKernel(unsigned long long* __restrict__ counters, unsigned long long* __restrict__ iterationTable, uint32_t iteration,
    Count count, const uint32_t* __restrict__ pos, const uint64_t* __restrict__ pairOffsets)
{
    if(threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long records = count.device ? *count.device : count.bound;     // what the hash kernel found (may exceed the capacity)
    const unsigned long long n = count.get();
    const unsigned long long buckets = (pos && n) ? pos[n] : 0, pairs = (pairOffsets && n) ? pairOffsets[n] : 0;
    counters[C_PAIRS] += pairs;
    counters[C_BUCKETS] = buckets;
    iterationTable[4ULL * iteration + 0] = records;
    iterationTable[4ULL * iteration + 1] = buckets;
    iterationTable[4ULL * iteration + 2] = counters[C_PAIRS];
    iterationTable[4ULL * iteration + 3] = 0;
    if(records > counters[C_MAX_RECORDS]) counters[C_MAX_RECORDS] = records;
    counters[C_RECORDS] = 0;
}

// The same for the records of ALL iterations in one sorted array (key = iteration << shift | bucket id): thread t files
// iteration t -- its records are [first record with key >= t << shift, first with key >= (t + 1) << shift), its buckets and its
// pair keys what the two scans say at those positions.  counters[C_RECORDS] holds what the hash kernel found in all.
__global__ void __launch_bounds__(64)
noteAllIterationsKernel(unsigned long long* __restrict__ counters, unsigned long long* __restrict__ iterationTable, uint32_t iterations,
    RecordKeys keys, Count count, const uint32_t* __restrict__ pos, const uint64_t* __restrict__ pairOffsets)
{
    const unsigned long long found = count.device ? *count.device : count.bound;
    const uint64_t n = count.get();
    auto firstAtLeast = [&](uint32_t t) -> uint64_t {           // first record of iteration >= t
        if(t >= iterations) return n;
        uint64_t lo = 0, hi = n;
        while(lo < hi) { const uint64_t mid = (lo + hi) >> 1; if(keys.iteration(mid) < t) lo = mid + 1; else hi = mid; }
        return lo;
    };
    for(uint32_t t = threadIdx.x; t < iterations; t += blockDim.x) {
        const uint64_t begin = firstAtLeast(t), end = firstAtLeast(t + 1);
        iterationTable[4ULL * t + 0] = end - begin;
        iterationTable[4ULL * t + 1] = n ? pos[end] - pos[begin] : 0;
        iterationTable[4ULL * t + 2] = n ? pairOffsets[end] : 0;
        iterationTable[4ULL * t + 3] = 0;
    }
    if(threadIdx.x == 0) {
        counters[C_PAIRS] += n ? pairOffsets[n] : 0;
        counters[C_BUCKETS] = n ? pos[n] : 0;
        counters[C_TOTAL_RECORDS] = found;
        counters[C_RECORDS] = 0;
    }
}

__global__ void __launch_bounds__(256)
fillKernel(uint32_t* __restrict__ p, uint64_t n, uint32_t value)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i < n) p[i] = value;
}

// K5 for ALL iterations at once (src/LowHash0.cpp:462-472, merge :493-562, summary :184-196).  The pair keys of
// every iteration, sorted by key with a STABLE sort: the occurrences of a pair are adjacent and in iteration order.
// The reference folds them iteration after iteration into a uint16_t frequency (src/LowHash0.hpp:116: additions
// wrap), and reports after each iteration how many pairs it holds (total) and how many have frequency >=
// minFrequency (high frequency).  The thread of a pair's first occurrence replays that history: after the pair's
// occurrences of iteration t, frequency = occurrences so far mod 2^16.  What it contributes to the two per-iteration
// counters are DIFFERENCES -- +1 to total at the pair's first iteration, +-1 to high frequency at every iteration
// where the pair crosses minFrequency -- accumulated per workgroup in LDS (one global atomic per workgroup, counter and
// iteration: workgroups are persistent); the host takes the running sums.
// flags[i] = 1 for the first occurrence of a pair whose final frequency makes it a candidate (:204-214).
constexpr int EVALUATE_LDS_ITERATIONS = 2048;
__global__ void __launch_bounds__(256)
evaluatePairsKernel(
    const uint64_t* __restrict__ keys, const uint32_t* __restrict__ tags, uint64_t n, uint32_t iterations, uint32_t minFrequency,
    unsigned long long* __restrict__ highDelta, unsigned long long* __restrict__ totalDelta, uint32_t* __restrict__ flags)
{
    __shared__ int sHigh[EVALUATE_LDS_ITERATIONS], sTotal[EVALUATE_LDS_ITERATIONS];
    const bool inLds = iterations <= uint32_t(EVALUATE_LDS_ITERATIONS);
    // A run has ten iterations or so: every thread's atomics would land on the same ten words (a wavefront's 64 on one word take
    // 64 turns in the LDS).  With few iterations each gets EVALUATE_COPIES words, one per bank, a lane adding to the one of its
    // number: two lanes to a word at most.
    constexpr uint32_t EVALUATE_COPIES = 32;
    const bool spread = iterations * EVALUATE_COPIES <= uint32_t(EVALUATE_LDS_ITERATIONS);
    const uint32_t slots = spread ? iterations * EVALUATE_COPIES : iterations;
    const uint32_t copy = threadIdx.x & (EVALUATE_COPIES - 1u);
    if(inLds) {
        for(uint32_t t = threadIdx.x; t < slots; t += blockDim.x) { sHigh[t] = 0; sTotal[t] = 0; }
        __syncthreads();
    }
    auto add = [&](int* lds, unsigned long long* global, uint32_t t, int delta) {
        if(inLds) atomicAdd(&lds[spread ? t * EVALUATE_COPIES + copy : t], delta);
        else atomicAdd(&global[t], (unsigned long long)(long long)delta);
    };
    for(uint64_t base = uint64_t(blockIdx.x) * blockDim.x; base <= n; base += uint64_t(gridDim.x) * blockDim.x) {
        const uint64_t i = base + threadIdx.x;
        if(i > n) continue;
        if(i == n) { flags[i] = 0u; continue; }
        const uint64_t key = keys[i];
        uint32_t flag = 0;
        if(i == 0 || keys[i - 1] != key) {
            uint32_t occurrences = 0;
            bool high = false, first = true;
            uint64_t at = i;
            while(at < n && keys[at] == key) {
                const uint32_t t = tags[at];
                while(at < n && keys[at] == key && tags[at] == t) { ++occurrences; ++at; }
                if(first) { add(sTotal, totalDelta, t, 1); first = false; }
                const bool now = (occurrences & 0xffffu) >= minFrequency;
                if(now != high) { add(sHigh, highDelta, t, now ? 1 : -1); high = now; }
            }
            flag = high ? 1u : 0u;
        }
        flags[i] = flag;
    }
    if(inLds) {
        __syncthreads();
        for(uint32_t t = threadIdx.x; t < iterations; t += blockDim.x) {
            int high = 0, total = 0;
            if(spread) for(uint32_t c = 0; c < EVALUATE_COPIES; c++) { high += sHigh[t * EVALUATE_COPIES + c]; total += sTotal[t * EVALUATE_COPIES + c]; }
            else { high = sHigh[t]; total = sTotal[t]; }
            if(high) atomicAdd(&highDelta[t], (unsigned long long)(long long)high);
            if(total) atomicAdd(&totalDelta[t], (unsigned long long)(long long)total);
        }
    }
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
    lowhashBuffers.reset();
    for(hipEvent_t e : alignEvents) (void)hipEventDestroy(e);
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
    // How often two markers of this read set are equal -- the random background of a candidate's marker-match matrix, nx ny times
    // this -- from a sample of 2^16 markers spread over all reads: floor(log2(1 / P)) - 1, never below 13 (k = 10 at markerDensity
    // 0.1 has about 7 900 marker k-mers; k = 14 has 640 000: sixty times fewer random matches per pair of reads).  The aligner sizes
    // its match lists and picks its cell-table classes by it (align4_prepare.hpp); a wrong estimate costs speed, never results --
    // whatever overflows its room or its table is detected on the device and runs again in a larger one.
    matchShift = 13;
    if(markerCount >= 2) {
        const uint32_t sampleCount = uint32_t(std::min<uint64_t>(markerCount, 65536));
        const uint64_t stride = markerCount / sampleCount;
        DeviceBuffer<uint32_t> sampleDevice;
        sampleDevice.reserve(sampleCount, stream);
        hipLaunchKernelGGL(sampleMarkersKernel, dim3(divUp(sampleCount, 256)), dim3(256), 0, stream, (const uint32_t*)kmerIds.data(), stride, sampleCount, sampleDevice.data());
        HIP_CHECK(hipGetLastError());
        std::vector<uint32_t> sample(sampleCount);
        HIP_CHECK(hipMemcpyAsync(sample.data(), sampleDevice.data(), sampleCount * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        std::sort(sample.begin(), sample.end());
        uint64_t equalPairs = 0;
        for(size_t i = 0; i < sample.size(); ) {
            size_t j = i + 1;
            while(j < sample.size() && sample[j] == sample[i]) ++j;
            equalPairs += uint64_t(j - i) * uint64_t(j - i - 1) / 2;
            i = j;
        }
        const double pairs = double(sampleCount) * double(sampleCount - 1) / 2.;
        int shift = 30;
        if(equalPairs) { shift = 0; while(shift < 30 && double(equalPairs) * double(2ULL << shift) <= pairs) ++shift; shift -= 1; }      // floor(log2(pairs / equalPairs)) - 1
        matchShift = std::min(std::max(shift, 13), 30);
        if(const char* e = std::getenv("SHASTA_MI355X_MATCH_SHIFT")) matchShift = std::min(std::max(std::atoi(e), 8), 30);       // (tests and timing experiments)
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

// The row of the kernel table a launch for this m is booked under: the template instance that runs, as a profiler prints it
// (M_FIXED, all iterations in one pass or not, the records' key type).
const char* hashKernelName(uint32_t m, bool all = false, bool wideKeys = false)
{
    if(all && wideKeys) {
        switch(m) {
            case 3: return "hashWindowsKernel<3, true, unsigned long>";
            case 4: return "hashWindowsKernel<4, true, unsigned long>";
            case 5: return "hashWindowsKernel<5, true, unsigned long>";
            default: return "hashWindowsKernel<0, true, unsigned long>";
        }
    }
    if(all) {
        switch(m) {
            case 3: return "hashWindowsKernel<3, true, unsigned int>";
            case 4: return "hashWindowsKernel<4, true, unsigned int>";
            case 5: return "hashWindowsKernel<5, true, unsigned int>";
            default: return "hashWindowsKernel<0, true, unsigned int>";
        }
    }
    switch(m) {
        case 3: return "hashWindowsKernel<3, false, unsigned int>";
        case 4: return "hashWindowsKernel<4, false, unsigned int>";
        case 5: return "hashWindowsKernel<5, false, unsigned int>";
        default: return "hashWindowsKernel<0, false, unsigned int>";
    }
}

// iterations = 0: one iteration with `seed`; otherwise all of them in one pass (hashWindowsKernel<m, true>).
template<class K = uint32_t>
void launchHash(Context& ctx, uint32_t m, uint64_t seed, uint64_t threshold, uint32_t mask,
    uint64_t markerBegin, uint64_t markerEnd,
    K* outKeys, uint64_t* outVals, unsigned long long* counter, uint64_t capacity, uint32_t iterations = 0, uint32_t iterationShift = 0)
{
    const uint64_t tiles = (markerEnd + HASH_TILE - 1) / HASH_TILE - markerBegin / HASH_TILE;
    if(tiles == 0) return;
    // Persistent wavefronts: 256 CUs x 8 blocks x 4 independent wavefronts, each walking many tiles, so that one global
    // atomic serves a few hundred low hashes.
    const unsigned blocks = unsigned(std::min<uint64_t>(divUp(tiles, uint64_t(HASH_THREADS / WAVE)), 256 * 8));
#define SHASTA_LAUNCH_HASH(MF, ALL) hipLaunchKernelGGL((hashWindowsKernel<MF, ALL, K>), dim3(blocks), dim3(HASH_THREADS), 0, ctx.stream, \
        (const uint32_t*)ctx.kmerIds.data(), (const uint64_t*)ctx.toc.data(), (const uint8_t*)ctx.readFlags.data(), \
        (const uint4*)ctx.tileDesc.data(), markerBegin, markerEnd, ctx.markerCount, \
        m, seed, threshold, mask, outKeys, outVals, counter, capacity, iterations, iterationShift)
    if(iterations) {
        switch(m) {
            case 3: SHASTA_LAUNCH_HASH(3, true); break;
            case 4: SHASTA_LAUNCH_HASH(4, true); break;
            case 5: SHASTA_LAUNCH_HASH(5, true); break;
            default: SHASTA_LAUNCH_HASH(0, true); break;
        }
    } else if constexpr (std::is_same<K, uint32_t>::value) {       // (one iteration's bucket ids always fit 32 bits)
        switch(m) {
            case 3: SHASTA_LAUNCH_HASH(3, false); break;
            case 4: SHASTA_LAUNCH_HASH(4, false); break;
            case 5: SHASTA_LAUNCH_HASH(5, false); break;
            default: SHASTA_LAUNCH_HASH(0, false); break;
        }
    } else {
        MI355X_ASSERT(iterations != 0);
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

// keys[i] |= owner << 56, owner = the rank whose bucket range [bounds[r], bounds[r + 1]) holds the key's bucket id (low word).
__global__ void __launch_bounds__(256)
tagOwnerKernel(uint64_t* __restrict__ keys, uint64_t n, const uint32_t* __restrict__ bounds, uint32_t world)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i >= n) return;
    const uint64_t key = keys[i];
    const uint32_t bucket = uint32_t(key);
    uint32_t owner = 0;
    for(uint32_t r = 1; r < world; r++) owner += bucket >= bounds[r] ? 1u : 0u;       // (bounds ascend; world is small)
    keys[i] = (key & 0x00ffffffffffffffULL) | (uint64_t(owner) << 56);
}

// ---------------------------------------------------------------------------
// The LowHash0 job: the state of one LowHash0::LowHash0 call.
//
// One GPU (lowhash0Run): per iteration the host only ENQUEUES -- hash, radix sort of the records, bucketKernel, a
// one-thread kernel that files the iteration's counters -- and after the last iteration reads the counters back once.
// The pair keys of all iterations accumulate in one array and are evaluated together (evaluatePairsKernel): one
// sort instead of a table merge per iteration.  Capacities (records per iteration, pair keys) are guesses remembered
// by the context; kernels never write past them, the counters stay exact, and a job whose guess was too small runs
// again with room for everything (first call on a context at most).  With minHashIterationCount = 0 the iteration
// control needs the high-frequency count after every iteration (src/LowHash0.cpp:137-149): then the accumulated keys
// are evaluated after every iteration.
//
// Several GPUs: the same kernels behind stage functions, with the two exchanges in the caller --
//   hash(iteration)      K1 on the rank's reads, records sorted by bucket id, split by bucket owner
//   buckets(records)     the records this rank owns -> statistics, histogram, this iteration's pair keys sorted and
//                        split by owner of readId0
//   merge(keys)          appends the keys this rank owns; evaluates them on request
//   finish()             candidates of the rank's readId0 range, statistics, per-iteration counters (partial sums)
// ---------------------------------------------------------------------------
struct LowHash0Job {
    shasta_lowhash0_params p;
    int rank = 0, world = 1;
    std::vector<uint64_t> boundaries;       // world + 1 read ids: rank r owns readId0 in [boundaries[r], boundaries[r+1])
    uint64_t log2BucketCount = 0, bucketCount = 0, hashThreshold = 0;
    uint32_t mask = 0, minFrequency = 0;
    int readBits = 0, pairKeyBits = 0;
    uint64_t markerBegin = 0, markerEnd = 0;
    uint64_t recCapacity = 0, pairCapacity = 0;
    uint64_t iterations = 0;                // iterations whose pair keys have been appended
    uint64_t pairCount = 0;                 // accumulated pair keys (host copy, valid after a read-back)
    uint64_t evaluatedIterations = ~0ULL, evaluatedPairs = ~0ULL, candidateCount = 0;
    bool pairsInB = false;                  // which side of the ping-pong holds the accumulated keys
    DeviceBuffer<uint32_t> recKeysA, recKeysB, pairTagsA, pairTagsB, flags, pos, starts, scanTemp32, boundKeys32, statKeysA, statKeysB;
    DeviceBuffer<uint64_t> recValsA, recValsB, pairKeysA, pairKeysB, iterKeysA, iterKeysB, pairCounts, scanTemp64, boundKeys64, boundOut;
    DeviceBuffer<unsigned long long> counters, stats, sizeHist, iterationTable, overflowSizes, highPerIteration, totalPerIteration;
    DeviceBuffer<shasta_oriented_read_pair> candidatesDevice;
    uint64_t histRows = 0;                  // rows (iterations) sizeHist / iterationTable / high / total have room for
    std::vector<size_t> hashHandles;        // kernel-table entries of the hash launches (bytes are filled in after the read-back)
    std::vector<uint64_t> highHost, totalHost;
    static constexpr uint32_t overflowCapacity = 1 << 20;

    uint64_t* pairKeys() { return pairsInB ? pairKeysB.data() : pairKeysA.data(); }
    uint32_t* pairTags() { return pairsInB ? pairTagsB.data() : pairTagsA.data(); }
    // The device allocations of a finished job (the context keeps it for that): a call does not allocate and free some thirty
    // buffers of hundreds of megabytes between its two events (that was most of the 5 ms between the kernels' 16 ms and the
    // call's 21, and an occasional 40 ms in the second call after a context was made).
    void scrambleBuffers(hipStream_t stream)
    {
        SharedCapacityMember* const all[] = {&recKeysA, &recKeysB, &pairTagsA, &pairTagsB, &flags, &pos, &starts, &scanTemp32, &boundKeys32, &statKeysA, &statKeysB,
            &recValsA, &recValsB, &pairKeysA, &pairKeysB, &iterKeysA, &iterKeysB, &pairCounts, &scanTemp64, &boundKeys64, &boundOut,
            &counters, &stats, &sizeHist, &iterationTable, &overflowSizes, &highPerIteration, &totalPerIteration, &candidatesDevice};
        for(SharedCapacityMember* b : all) b->scramble(stream);
    }
    void adoptBuffers(LowHash0Job& old)
    {
        recKeysA.swap(old.recKeysA); recKeysB.swap(old.recKeysB); pairTagsA.swap(old.pairTagsA); pairTagsB.swap(old.pairTagsB);
        flags.swap(old.flags); pos.swap(old.pos); starts.swap(old.starts); scanTemp32.swap(old.scanTemp32); boundKeys32.swap(old.boundKeys32);
        statKeysA.swap(old.statKeysA); statKeysB.swap(old.statKeysB);      // (round 6: these two were left out -- allocated and freed by every call, SHASTA_MI355X_LOG_ALLOC=1 on the emulated build showed it)
        recValsA.swap(old.recValsA); recValsB.swap(old.recValsB); pairKeysA.swap(old.pairKeysA); pairKeysB.swap(old.pairKeysB);
        iterKeysA.swap(old.iterKeysA); iterKeysB.swap(old.iterKeysB); pairCounts.swap(old.pairCounts); scanTemp64.swap(old.scanTemp64);
        boundKeys64.swap(old.boundKeys64); boundOut.swap(old.boundOut);
        counters.swap(old.counters); stats.swap(old.stats); sizeHist.swap(old.sizeHist); iterationTable.swap(old.iterationTable);
        overflowSizes.swap(old.overflowSizes); highPerIteration.swap(old.highPerIteration); totalPerIteration.swap(old.totalPerIteration);
        candidatesDevice.swap(old.candidatesDevice);
    }
};

// The job is over: its allocations stay with the context for the next one.
void retireJob(Context& ctx)
{
    if(ctx.lowhashJob) ctx.lowhashBuffers = ctx.lowhashJob;
    ctx.lowhashJob.reset();
}

namespace {
LowHash0Job& jobOf(Context& ctx)
{
    if(!ctx.lowhashJob) throw std::runtime_error("LowHash0: no job in progress (call begin first).");
    return *static_cast<LowHash0Job*>(ctx.lowhashJob.get());
}

// Room for `rows` iterations in the per-iteration tables (kept when growing: the dynamic iteration control adds rows).
void reserveIterationRows(LowHash0Job& job, uint64_t rows, hipStream_t stream)
{
    if(rows <= job.histRows) return;
    const uint64_t newRows = std::max<uint64_t>(rows, 2 * job.histRows);
    if(job.histRows == 0 && job.sizeHist.capacity() >= newRows * SIZE_HIST_CAP && job.iterationTable.capacity() >= newRows * 4) {
        // A new job in the allocations of the previous one: they only need clearing.
        HIP_CHECK(hipMemsetAsync(job.sizeHist.data(), 0, newRows * SIZE_HIST_CAP * sizeof(unsigned long long), stream));
        HIP_CHECK(hipMemsetAsync(job.iterationTable.data(), 0, newRows * 4 * sizeof(unsigned long long), stream));
        job.highPerIteration.reserve(newRows, stream); job.totalPerIteration.reserve(newRows, stream);
        job.histRows = newRows;
        return;
    }
    DeviceBuffer<unsigned long long> hist, table;
    hist.reserve(newRows * SIZE_HIST_CAP, stream); table.reserve(newRows * 4, stream);
    HIP_CHECK(hipMemsetAsync(hist.data(), 0, newRows * SIZE_HIST_CAP * sizeof(unsigned long long), stream));
    HIP_CHECK(hipMemsetAsync(table.data(), 0, newRows * 4 * sizeof(unsigned long long), stream));
    if(job.histRows) {
        HIP_CHECK(hipMemcpyAsync(hist.data(), job.sizeHist.data(), job.histRows * SIZE_HIST_CAP * sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(table.data(), job.iterationTable.data(), job.histRows * 4 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream));
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    job.sizeHist.swap(hist); job.iterationTable.swap(table);
    job.highPerIteration.reserve(newRows, stream); job.totalPerIteration.reserve(newRows, stream);
    job.histRows = newRows;
}

void reservePairs(LowHash0Job& job, uint64_t capacity, hipStream_t stream, bool keep)
{
    job.pairKeysA.reserve(capacity, stream, keep); job.pairKeysB.reserve(capacity, stream, keep);
    job.pairTagsA.reserve(capacity, stream, keep); job.pairTagsB.reserve(capacity, stream, keep);
    job.pairCapacity = std::min<uint64_t>(std::min(job.pairKeysA.capacity(), job.pairKeysB.capacity()), std::min(job.pairTagsA.capacity(), job.pairTagsB.capacity()));
}

// K1 of one iteration into (recKeysA, recValsA); the record count stays on the device (counters[C_RECORDS]).
void enqueueHash(Context& ctx, LowHash0Job& job, uint64_t iteration)
{
    hipStream_t stream = ctx.stream;
    job.recKeysA.reserve(job.recCapacity, stream); job.recKeysB.reserve(job.recCapacity, stream);
    job.recValsA.reserve(job.recCapacity, stream); job.recValsB.reserve(job.recCapacity, stream);
    const KernelTimers::Span span = ctx.timers.begin(hashKernelName(uint32_t(job.p.m)), stream);
    launchHash(ctx, uint32_t(job.p.m), iteration * 37, job.hashThreshold, job.mask, job.markerBegin, job.markerEnd,
        job.recKeysA.data(), job.recValsA.data(), job.counters.data() + C_RECORDS, job.recCapacity);
    job.hashHandles.push_back(ctx.timers.end(span, 4 * (job.markerEnd - job.markerBegin), job.markerEnd - job.markerBegin));
}

// K2: the records sorted by bucket id (radix partition on the bucket id).  Which side holds the result depends on
// the key width only.  wideKeys (all iterations in one pass where iteration | bucket id does not fit 32 bits): the key
// buffers hold 64-bit keys, iteration << 32 | bucket id; `keys` is returned as their 32-bit words.
void enqueueSortRecords(Context& ctx, LowHash0Job& job, const uint32_t*& keys, const uint64_t*& vals, Count count, uint32_t allIterations = 0, bool wideKeys = false)
{
    hipStream_t stream = ctx.stream;
    keys = job.recKeysA.data(); vals = job.recValsA.data();
    const int bits = wideKeys ? 32 + bitsFor(allIterations - 1)
        : int(job.log2BucketCount) + (allIterations ? bitsFor(allIterations - 1) : 0);      // (allIterations: the iteration above the bucket id)
    const uint64_t passes = (uint64_t(bits) + 7) / 8;
    const KernelTimers::Span span = ctx.timers.begin(allIterations ? "radix sort of the low-hash records of all iterations" : "radix sort of low-hash records", stream);
    const bool inB = wideKeys
        ? radixSort<uint64_t, uint64_t, true>(reinterpret_cast<uint64_t*>(job.recKeysA.data()), reinterpret_cast<uint64_t*>(job.recKeysB.data()),
            job.recValsA.data(), job.recValsB.data(), count, bits, ctx.sortWs, stream)
        : radixSort<uint32_t, uint64_t, true>(job.recKeysA.data(), job.recKeysB.data(), job.recValsA.data(), job.recValsB.data(), count, bits, ctx.sortWs, stream);
    if(inB) { keys = job.recKeysB.data(); vals = job.recValsB.data(); }
    // 12 (16) bytes per record read + written per 8-bit pass; the record count is booked from the expected fraction.
    const uint64_t expected = uint64_t(std::min(std::max(job.p.hashFraction, 0.), 1.) * double(job.markerEnd - job.markerBegin)) * std::max<uint32_t>(1, allIterations);
    (void)ctx.timers.end(span, 2 * (wideKeys ? 16 : 12) * expected * passes, expected);
}

// K3 + K4 on sorted records: statistics, histogram row `iteration`, pair keys appended at counters[C_PAIRS]; then the
// iteration's counters are filed (noteIterationKernel).
// allIterations > 0: the records of that many iterations in one array, key = iteration << log2BucketCount | bucket id (`iteration` unused).
bool statisticsByAtomics() { const char* e = std::getenv("SHASTA_MI355X_STATISTICS_ATOMICS"); return e && std::atoi(e) != 0; }
void enqueueBuckets(Context& ctx, LowHash0Job& job, const uint32_t* keyWords, const uint64_t* vals, Count count, uint64_t iteration,
    uint64_t* pairKeys, uint32_t* pairTags, uint64_t pairCapacity, uint32_t allIterations = 0, bool wideKeys = false)
{
    hipStream_t stream = ctx.stream;
    const uint64_t bound = count.bound;
    unsigned long long* counters = job.counters.data();
    const RecordKeys keys{keyWords, wideKeys ? 2u : 1u, wideKeys ? 0u : uint32_t(job.log2BucketCount)};
    const RecordKeys iterationKeys = allIterations ? keys : RecordKeys{nullptr, 1u, 0u};
    if(bound == 0 && allIterations) {
        hipLaunchKernelGGL(noteAllIterationsKernel, dim3(1), dim3(64), 0, stream, counters, job.iterationTable.data(), allIterations,
            keys, count, (const uint32_t*)nullptr, (const uint64_t*)nullptr);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if(bound == 0) {
        hipLaunchKernelGGL(noteIterationKernel, dim3(1), dim3(64), 0, stream, counters, job.iterationTable.data(), uint32_t(iteration), count, (const uint32_t*)nullptr, (const uint64_t*)nullptr);
        HIP_CHECK(hipGetLastError());
        return;
    }
    const uint64_t expected = uint64_t(std::min(std::max(job.p.hashFraction, 0.), 1.) * double(job.markerEnd - job.markerBegin)) * std::max<uint32_t>(1, allIterations);
    job.flags.reserve(bound + 1, stream); job.pos.reserve(bound + 1, stream); job.starts.reserve(bound + 2, stream);
    job.scanTemp32.reserve(scanTempElements(bound + 1), stream);
    job.pairCounts.reserve(bound + 1, stream); job.scanTemp64.reserve(scanTempElements(bound + 1), stream);
    const unsigned g = divUp(bound + 1, 256);
    SHASTA_TIMED(ctx, "bucket boundaries (heads, scan, starts)", stream, 12 * expected, expected,
        hipLaunchKernelGGL(markHeadsKernel, dim3(g), dim3(256), 0, stream, keys, count, job.flags.data());
        exclusiveScan<uint32_t>(job.flags.data(), job.pos.data(), bound + 1, job.scanTemp32.data(), stream);
        hipLaunchKernelGGL(groupStartsKernel, dim3(g), dim3(256), 0, stream, (const uint32_t*)job.flags.data(), (const uint32_t*)job.pos.data(), count, job.starts.data()));
    // The per-read statistics through keys, a partition pass and LDS counts (readStatisticsKernel) wherever 3 R fits the 32-bit key;
    // SHASTA_MI355X_STATISTICS_ATOMICS=1: an atomic per record, as until round 3 (the A/B switch).
    const uint64_t statEntries = 3 * ctx.readCount;
    const bool statsByKeys = statEntries < (1ULL << 32) && !statisticsByAtomics();
    if(statsByKeys) { job.statKeysA.reserve(bound + 1, stream); job.statKeysB.reserve(bound + 1, stream); }
    SHASTA_TIMED(ctx, "bucketStatsKernel + scan of pair counts", stream, 12 * expected, expected,
        hipLaunchKernelGGL(bucketStatsKernel, dim3(g), dim3(256), 0, stream,
            vals, (const uint32_t*)job.pos.data(), (const uint32_t*)job.starts.data(), count,
            job.p.minBucketSize, job.p.maxBucketSize, uint32_t(iteration), iterationKeys, job.stats.data(),
            job.sizeHist.data() + (allIterations ? 0 : iteration * SIZE_HIST_CAP),
            job.overflowSizes.data(), LowHash0Job::overflowCapacity, counters, job.pairCounts.data(), statsByKeys ? job.statKeysA.data() : (uint32_t*)nullptr);
        exclusiveScan<uint64_t>(job.pairCounts.data(), job.pairCounts.data(), bound + 1, job.scanTemp64.data(), stream));
    if(statsByKeys) {
        int keyBits = 1;
        while(keyBits < 32 && (1ULL << keyBits) < statEntries) ++keyBits;
        const int partitionBits = std::max(0, keyBits - STAT_LOW_BITS);
        // Booked: 4 bytes per record read by the counts, 12 per record and pass of the partition (histogram read, scatter read + write).
        SHASTA_TIMED(ctx, "readStatisticsKernel + partition of the statistics keys", stream, (4 + 12 * uint64_t((partitionBits + 7) / 8)) * expected, expected,
            const bool inB = radixSort<uint32_t, uint32_t, false>(job.statKeysA.data(), job.statKeysB.data(), nullptr, nullptr, count, partitionBits, ctx.sortWs, stream, STAT_LOW_BITS);
            hipLaunchKernelGGL(readStatisticsKernel, dim3(divUp(bound, STAT_SPAN)), dim3(256), 0, stream,
                (const uint32_t*)(inB ? job.statKeysB.data() : job.statKeysA.data()), count, job.stats.data(), statEntries));
    }
    SHASTA_TIMED(ctx, "pairWriteKernel", stream, 0, expected,
        hipLaunchKernelGGL(pairWriteKernel, dim3(divUp(bound, 256)), dim3(256), 0, stream,
            vals, (const uint32_t*)job.pos.data(), (const uint32_t*)job.starts.data(), (const uint64_t*)job.pairCounts.data(), count, job.readBits, uint32_t(iteration),
            iterationKeys, (const unsigned long long*)counters, pairKeys, pairTags, pairCapacity));
    if(allIterations) hipLaunchKernelGGL(noteAllIterationsKernel, dim3(1), dim3(64), 0, stream, counters, job.iterationTable.data(), allIterations,
        keys, count, (const uint32_t*)job.pos.data(), (const uint64_t*)job.pairCounts.data());
    else hipLaunchKernelGGL(noteIterationKernel, dim3(1), dim3(64), 0, stream, counters, job.iterationTable.data(), uint32_t(iteration), count,
        (const uint32_t*)job.pos.data(), (const uint64_t*)job.pairCounts.data());
    HIP_CHECK(hipGetLastError());
}

// K5 over everything accumulated so far: sorts the pairCount keys, fills highPerIteration / totalPerIteration
// [0, iterations) and the candidate flags + their scan.  Ends with ONE read-back (candidate count, both arrays).
void evaluate(Context& ctx, LowHash0Job& job)
{
    hipStream_t stream = ctx.stream;
    const uint64_t n = job.pairCount, iterations = job.iterations;
    if(job.evaluatedIterations == iterations && job.evaluatedPairs == n) return;
    MI355X_ASSERT(n < (1ULL << 32) - 1 && iterations < (1ULL << 32));
    reserveIterationRows(job, std::max<uint64_t>(1, iterations), stream);
    HIP_CHECK(hipMemsetAsync(job.highPerIteration.data(), 0, std::max<uint64_t>(1, iterations) * sizeof(unsigned long long), stream));
    HIP_CHECK(hipMemsetAsync(job.totalPerIteration.data(), 0, std::max<uint64_t>(1, iterations) * sizeof(unsigned long long), stream));
    job.candidateCount = 0;
    job.highHost.assign(iterations, 0); job.totalHost.assign(iterations, 0);
    if(n) {
        {
            const KernelTimers::Span span = ctx.timers.begin("radix sort of the pair keys of all iterations", stream);
            uint64_t* ka = job.pairKeys(); uint64_t* kb = job.pairsInB ? job.pairKeysA.data() : job.pairKeysB.data();
            uint32_t* ta = job.pairTags(); uint32_t* tb = job.pairsInB ? job.pairTagsA.data() : job.pairTagsB.data();
            if(radixSort<uint64_t, uint32_t, true>(ka, kb, ta, tb, n, job.pairKeyBits, ctx.sortWs, stream)) job.pairsInB = !job.pairsInB;
            (void)ctx.timers.end(span, 2 * 12 * n * uint64_t((job.pairKeyBits + 7) / 8), n);
        }
        job.flags.reserve(n + 1, stream); job.pos.reserve(n + 1, stream);
        job.scanTemp32.reserve(scanTempElements(n + 1), stream);
        SHASTA_TIMED(ctx, "evaluatePairsKernel + scan of candidate flags", stream, 12 * n, n,
            hipLaunchKernelGGL(evaluatePairsKernel, dim3(std::min<unsigned>(divUp(n + 1, 256), 2048)), dim3(256), 0, stream,
                (const uint64_t*)job.pairKeys(), (const uint32_t*)job.pairTags(), n, uint32_t(iterations), job.minFrequency,
                job.highPerIteration.data(), job.totalPerIteration.data(), job.flags.data());
            exclusiveScan<uint32_t>(job.flags.data(), job.pos.data(), n + 1, job.scanTemp32.data(), stream));
        HIP_CHECK(hipGetLastError());
        uint32_t candidates = 0;
        HIP_CHECK(hipMemcpyAsync(&candidates, job.pos.data() + n, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        if(iterations) {
            HIP_CHECK(hipMemcpyAsync(job.highHost.data(), job.highPerIteration.data(), iterations * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipMemcpyAsync(job.totalHost.data(), job.totalPerIteration.data(), iterations * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        }
        HIP_CHECK(hipStreamSynchronize(stream));
        job.candidateCount = candidates;
        // The kernel left differences: running sums (two's complement arithmetic carries the -1s).
        for(uint64_t t = 1; t < iterations; t++) { job.highHost[t] += job.highHost[t - 1]; job.totalHost[t] += job.totalHost[t - 1]; }
    }
    job.evaluatedIterations = iterations; job.evaluatedPairs = n;
}

// Histogram rows (src/LowHash0.cpp:586-595) of one iteration from the (summed) size histogram.
void appendHistogramRows(uint64_t iteration, uint64_t bucketCount, uint64_t bucketsUsed,
    const uint64_t* sizeHistogram, const std::vector<uint32_t>& overflow, std::vector<uint64_t>& histogramRows)
{
    std::map<uint64_t, uint64_t> rows;
    if(bucketCount > bucketsUsed) rows[0] = bucketCount - bucketsUsed;
    for(int s = 1; s < SIZE_HIST_CAP; s++) if(sizeHistogram[s]) rows[uint64_t(s)] = sizeHistogram[s];
    for(uint32_t s : overflow) ++rows[s];
    for(const auto& r : rows) { histogramRows.push_back(iteration); histogramRows.push_back(r.first); histogramRows.push_back(r.second); }
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
    if(ctx.lowhashBuffers) { job.adoptBuffers(*static_cast<LowHash0Job*>(ctx.lowhashBuffers.get())); ctx.lowhashBuffers.reset(); }
    // (SHASTA_MI355X_SCRAMBLE=1, a test switch: what the last job left in the buffers this one takes over is overwritten with pseudo-random data)
    if(const char* e = std::getenv("SHASTA_MI355X_SCRAMBLE")) if(e[0] == '1') { job.scrambleBuffers(ctx.stream); HIP_CHECK(hipStreamSynchronize(ctx.stream)); }
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
    // First guesses (grown, and the job repeated, when they turn out too small; remembered by the context); the
    // fraction is clamped because any double is a legal hashFraction in the reference (>= 1 keeps nothing, < 0 nearly everything).
    const double expectedFraction = !(p.hashFraction > 0.) ? 0. : std::min(p.hashFraction, 1.);
    job.recCapacity = std::max<uint64_t>(1 << 16, uint64_t(2.0 * expectedFraction * double(job.markerEnd - job.markerBegin)) + (1 << 16));
    job.recCapacity = std::max(job.recCapacity, ctx.lowhashRecordsHint);
    const uint64_t plannedIterations = p.minHashIterationCount ? p.minHashIterationCount : 8;
    reservePairs(job, std::max<uint64_t>(std::max<uint64_t>(1 << 20, plannedIterations * job.recCapacity / 2), ctx.lowhashPairsHint), stream, false);

    job.counters.reserve(C_COUNT, stream);
    job.stats.reserve(3 * readCount, stream);
    job.overflowSizes.reserve(LowHash0Job::overflowCapacity, stream);
    job.boundKeys32.reserve(size_t(world) + 1, stream); job.boundKeys64.reserve(size_t(world) + 1, stream);
    job.boundOut.reserve(size_t(world) + 1, stream);
    HIP_CHECK(hipMemsetAsync(job.counters.data(), 0, C_COUNT * sizeof(unsigned long long), stream));
    HIP_CHECK(hipMemsetAsync(job.stats.data(), 0, 3 * readCount * sizeof(unsigned long long), stream));
    reserveIterationRows(job, std::max<uint64_t>(1, plannedIterations), stream);
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
    unsigned long long* counter = job.counters.data() + C_RECORDS;
    uint64_t n = 0;
    for(;;) {
        HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), stream));
        enqueueHash(ctx, job, iteration);
        n = readDevice(counter, stream);
        ctx.timers.amend(job.hashHandles.back(), 4 * (job.markerEnd - job.markerBegin) + 12 * std::min(n, job.recCapacity), job.markerEnd - job.markerBegin);
        if(n <= job.recCapacity) break;
        job.recCapacity = n + n / 4;           // estimate was too small: grow and redo this iteration
        ctx.lowhashRecordsHint = job.recCapacity;
    }
    HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), stream));
    MI355X_ASSERT(n < (1ULL << 32) - 1);
    const uint32_t* keys = nullptr; const uint64_t* vals = nullptr;
    enqueueSortRecords(ctx, job, keys, vals, Count(n));
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

// Stage 2.  (keys, vals): the n records of the buckets this rank owns (device pointers; any order when world > 1).
// Produces this iteration's pair keys, sorted and split by owner of readId0.
void lowhash0Buckets(Context& ctx, const uint32_t* keysIn, const uint64_t* valsIn, uint64_t n,
    uint64_t* sendOffsets, const uint64_t** pairKeysOut,
    uint64_t* bucketsUsedOut, uint64_t* sizeHistogramOut /*SIZE_HIST_CAP*/, std::vector<uint32_t>& overflowOut)
{
    LowHash0Job& job = jobOf(ctx);
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;
    MI355X_ASSERT(n < (1ULL << 32) - 1);
    const uint64_t iteration = job.iterations;
    reserveIterationRows(job, iteration + 1, stream);
    const uint32_t* keys = keysIn; const uint64_t* vals = valsIn;
    if(job.world > 1 && n) {
        // Concatenation of one sorted run per sender: sort again.
        job.recKeysA.reserve(n, stream); job.recKeysB.reserve(n, stream); job.recValsA.reserve(n, stream); job.recValsB.reserve(n, stream);
        if(keysIn != job.recKeysA.data()) HIP_CHECK(hipMemcpyAsync(job.recKeysA.data(), keysIn, n * 4, hipMemcpyDeviceToDevice, stream));
        if(valsIn != job.recValsA.data()) HIP_CHECK(hipMemcpyAsync(job.recValsA.data(), valsIn, n * 8, hipMemcpyDeviceToDevice, stream));
        enqueueSortRecords(ctx, job, keys, vals, Count(n));
    }
    // This iteration's pair keys go to a buffer of their own (they leave for their owners before they are appended),
    // sized for the most a set of n records can produce: every record pairs with fewer than min(maxBucketSize, n) others.
    unsigned long long* counters = job.counters.data();
    const uint64_t perRecord = std::min<uint64_t>(std::min<uint64_t>(job.p.maxBucketSize, 1ULL << 31), n);
    const uint64_t capacity = n * perRecord / 2 + 64;
    MI355X_ASSERT(capacity < (1ULL << 32) - 1);
    job.iterKeysA.reserve(capacity, stream); job.iterKeysB.reserve(capacity, stream);
    const uint64_t overflowBefore = readDevice(counters + C_OVERFLOW, stream);
    HIP_CHECK(hipMemsetAsync(counters + C_PAIRS, 0, sizeof(unsigned long long), stream));
    HIP_CHECK(hipMemsetAsync(counters + C_BUCKETS, 0, sizeof(unsigned long long), stream));
    enqueueBuckets(ctx, job, keys, vals, Count(n), iteration, job.iterKeysA.data(), nullptr, capacity);
    const uint64_t pairCount = readDevice(counters + C_PAIRS, stream);
    const uint64_t bucketsUsed = readDevice(counters + C_BUCKETS, stream);
    MI355X_ASSERT(pairCount <= capacity);
    *bucketsUsedOut = bucketsUsed;
    HIP_CHECK(hipMemcpyAsync(sizeHistogramOut, job.sizeHist.data() + iteration * SIZE_HIST_CAP, SIZE_HIST_CAP * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    const uint64_t overflowAfter = readDevice(counters + C_OVERFLOW, stream);
    if(overflowAfter > LowHash0Job::overflowCapacity) throw std::runtime_error("LowHash0: bucket-size overflow list exhausted.");
    overflowOut.clear();
    if(overflowAfter > overflowBefore) {
        std::vector<unsigned long long> entries(overflowAfter - overflowBefore);
        HIP_CHECK(hipMemcpyAsync(entries.data(), job.overflowSizes.data() + overflowBefore, entries.size() * 8, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        for(unsigned long long e : entries) overflowOut.push_back(uint32_t(e));
    }
    const uint64_t* pk = job.iterKeysA.data();
    if(pairCount) {
        MI355X_ASSERT(pairCount < (1ULL << 32) - 1);
        const KernelTimers::Span span = ctx.timers.begin("radix sort of pair keys (one iteration, staged)", stream);
        if(radixSort<uint64_t, uint32_t, false>(job.iterKeysA.data(), job.iterKeysB.data(), nullptr, nullptr, pairCount, job.pairKeyBits, ctx.sortWs, stream)) pk = job.iterKeysB.data();
        (void)ctx.timers.end(span, 2 * 8 * pairCount * uint64_t((job.pairKeyBits + 7) / 8), pairCount);
    }
    if(job.world == 1 || pairCount == 0) {
        for(int r = 0; r <= job.world; r++) sendOffsets[r] = (r == job.world) ? pairCount : 0;
    } else {
        hipLaunchKernelGGL(lowerBoundsKernel<uint64_t>, dim3(divUp(uint64_t(job.world) + 1, 64)), dim3(64), 0, stream,
            pk, pairCount, (const uint64_t*)job.boundKeys64.data(), uint32_t(job.world + 1), job.boundOut.data());
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(sendOffsets, job.boundOut.data(), (size_t(job.world) + 1) * 8, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        sendOffsets[0] = 0; sendOffsets[job.world] = pairCount;
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    *pairKeysOut = pk;
}

// Stage 3.  keys: the n pair keys of this iteration whose readId0 this rank owns (device pointer).  Appends them;
// with evaluateNow also evaluates everything appended so far and returns this rank's share of the latest iteration's
// "high frequency" and "total" counters (the dynamic iteration control needs them after every iteration).
void lowhash0Merge(Context& ctx, const uint64_t* keys, uint64_t n, bool evaluateNow, uint64_t* highFrequencyOut, uint64_t* totalOut)
{
    LowHash0Job& job = jobOf(ctx);
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;
    if(n) {
        const uint64_t needed = job.pairCount + n;
        if(needed > job.pairCapacity) reservePairs(job, needed + needed / 2, stream, true);
        HIP_CHECK(hipMemcpyAsync(job.pairKeys() + job.pairCount, keys, n * 8, hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(fillKernel, dim3(divUp(n, 256)), dim3(256), 0, stream, job.pairTags() + job.pairCount, n, uint32_t(job.iterations));
        HIP_CHECK(hipGetLastError());
        job.pairCount = needed;
    }
    ++job.iterations;
    *highFrequencyOut = 0; *totalOut = 0;
    if(evaluateNow) {
        evaluate(ctx, job);
        *highFrequencyOut = job.highHost.back(); *totalOut = job.totalHost.back();
    } else {
        HIP_CHECK(hipStreamSynchronize(stream));          // `keys` may be reused by the caller
    }
}

// ---- the staged job with all iterations in one pass (fixed minHashIterationCount): three calls and two exchanges per JOB
// instead of per iteration.  Keys are 64-bit: owner << 56 | iteration << 32 | bucket id.

// Whether the job that lowhash0Begin set up on this context can take all its iterations in one pass: a fixed number of them, at
// most 4096, at most 256 ranks, and the records of ALL iterations -- this rank's own and, the buckets being dealt evenly, about
// as many received -- within the 32-bit positions of one sort, with a factor of two to spare (the capacity is an estimate that
// grows when it was short).  Every rank of a job asks its own context and the job takes the one-pass form only if ALL say yes
// (Group::lowhash0Run, shasta_amd/distributed.py): otherwise iteration after iteration, which only needs ONE iteration's records
// to fit.  (Human-scale marker counts on few ranks, or hundreds of iterations, are where the answer is no.)
// (SHASTA_MI355X_ONE_PASS_RECORD_LIMIT lowers the limit: tests of the fall-back at sizes a test can hold;
// SHASTA_MI355X_DEBUG_ONE_PASS=1 says on stderr when a job falls back.)
uint64_t onePassRecordLimit()
{
    static const uint64_t limit = [] { const char* e = std::getenv("SHASTA_MI355X_ONE_PASS_RECORD_LIMIT"); return e ? std::min<uint64_t>(std::strtoull(e, nullptr, 10), (1ULL << 32) - 1) : (1ULL << 32) - 1; }();
    return limit;
}
void noteOnePassFallback(uint64_t iterations, uint64_t recCapacity)
{
    static const bool debug = [] { const char* e = std::getenv("SHASTA_MI355X_DEBUG_ONE_PASS"); return e && e[0] == '1'; }();
    if(debug) std::fprintf(stderr, "LowHash0: %llu iterations x %llu records do not fit one sort: iteration after iteration\n", (unsigned long long)iterations, (unsigned long long)recCapacity);
}
bool lowhash0OnePassFits(Context& ctx)
{
    const LowHash0Job& job = jobOf(ctx);
    const uint64_t I = job.p.minHashIterationCount;
    const bool fits = I >= 1 && I <= 4096 && job.world <= 256 && 2 * I * job.recCapacity < onePassRecordLimit();
    if(!fits && I >= 1) noteOnePassFallback(I, job.recCapacity);
    return fits;
}

// Stage 1 (all iterations).  sendOffsets[r..r+1] delimit the records of the buckets rank r owns in (*keys, *vals).
void lowhash0HashAll(Context& ctx, uint64_t* sendOffsets, const uint64_t** keysOut, const uint64_t** valsOut)
{
    LowHash0Job& job = jobOf(ctx);
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;
    const uint64_t I = job.p.minHashIterationCount;
    if(I < 1 || I > 4096) throw std::runtime_error("LowHash0: all iterations in one pass needs 1 <= minHashIterationCount <= 4096.");
    if(job.world > 256) throw std::runtime_error("LowHash0: at most 256 ranks.");
    unsigned long long* counter = job.counters.data() + C_RECORDS;
    uint64_t n = 0, capacity = 0;
    for(;;) {
        capacity = I * job.recCapacity;
        if(capacity >= (1ULL << 32) - 1) throw std::runtime_error("LowHash0: the low hashes of all " + std::to_string(I) + " iterations (" + std::to_string(capacity) +
            " records on this rank) exceed the 2^32 positions of one sort: run the job iteration after iteration (shasta_mi355x_lh_one_pass_fits says beforehand).");
        job.recKeysA.reserve(2 * capacity, stream); job.recKeysB.reserve(2 * capacity, stream);
        job.recValsA.reserve(capacity, stream); job.recValsB.reserve(capacity, stream);
        HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), stream));
        const KernelTimers::Span span = ctx.timers.begin(hashKernelName(uint32_t(job.p.m), true, true), stream);
        launchHash<uint64_t>(ctx, uint32_t(job.p.m), 0, job.hashThreshold, job.mask, job.markerBegin, job.markerEnd,
            reinterpret_cast<uint64_t*>(job.recKeysA.data()), job.recValsA.data(), counter, capacity, uint32_t(I), 32u);
        job.hashHandles.push_back(ctx.timers.end(span, 4 * (job.markerEnd - job.markerBegin), (job.markerEnd - job.markerBegin) * I));
        n = readDevice(counter, stream);
        ctx.timers.amend(job.hashHandles.back(), 4 * (job.markerEnd - job.markerBegin) + 16 * std::min(n, capacity), (job.markerEnd - job.markerBegin) * I);
        if(n <= capacity) break;
        job.recCapacity = (n + I - 1) / I + n / (4 * I) + 1;       // estimate was too small: grow and hash again
        ctx.lowhashRecordsHint = job.recCapacity;
    }
    HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), stream));
    const uint64_t* keys = reinterpret_cast<const uint64_t*>(job.recKeysA.data());
    const uint64_t* vals = job.recValsA.data();
    if(job.world == 1 || n == 0) {
        for(int r = 0; r <= job.world; r++) sendOffsets[r] = (r == job.world) ? n : 0;
    } else {
        // Owners into the keys' top byte, ONE stable radix pass on it (a sender only has to make each owner's records
        // contiguous: the receiver sorts them), split points by binary search.
        hipLaunchKernelGGL(tagOwnerKernel, dim3(divUp(n, 256)), dim3(256), 0, stream,
            reinterpret_cast<uint64_t*>(job.recKeysA.data()), n, (const uint32_t*)job.boundKeys32.data(), uint32_t(job.world));
        HIP_CHECK(hipGetLastError());
        const KernelTimers::Span span = ctx.timers.begin("records of all iterations partitioned by owner", stream);
        if(radixSort<uint64_t, uint64_t, true>(reinterpret_cast<uint64_t*>(job.recKeysA.data()), reinterpret_cast<uint64_t*>(job.recKeysB.data()),
            job.recValsA.data(), job.recValsB.data(), Count(n), 8, ctx.sortWs, stream, 56)) {
            keys = reinterpret_cast<const uint64_t*>(job.recKeysB.data()); vals = job.recValsB.data();
        }
        (void)ctx.timers.end(span, 2 * 16 * n, n);
        std::vector<uint64_t> bounds(size_t(job.world) + 1);
        for(int r = 0; r <= job.world; r++) bounds[r] = uint64_t(std::min(r, 255)) << 56;
        HIP_CHECK(hipMemcpyAsync(job.boundKeys64.data(), bounds.data(), bounds.size() * 8, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(lowerBoundsKernel<uint64_t>, dim3(divUp(uint64_t(job.world) + 1, 64)), dim3(64), 0, stream,
            keys, n, (const uint64_t*)job.boundKeys64.data(), uint32_t(job.world + 1), job.boundOut.data());
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(sendOffsets, job.boundOut.data(), (size_t(job.world) + 1) * 8, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        sendOffsets[0] = 0; sendOffsets[job.world] = n;
        // (boundKeys64 holds the pair keys' owner bounds for stage 2: put them back.)
        std::vector<uint64_t> b64(size_t(job.world) + 1);
        for(int r = 0; r <= job.world; r++) b64[r] = job.boundaries[r] << (job.readBits + 1);
        HIP_CHECK(hipMemcpyAsync(job.boundKeys64.data(), b64.data(), b64.size() * 8, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    *keysOut = keys; *valsOut = vals;
}

// Stage 2 (all iterations).  (keys, vals): the n records of the buckets this rank owns, of every iteration (device pointers, any
// order).  Statistics, histograms (rows of every iteration), the pair keys of all iterations with their iteration tags, sorted
// by key and split by owner of readId0.  bucketsUsedOut[iterations], sizeHistogramOut[iterations][SIZE_HIST_CAP];
// overflowOut: iteration << 32 | size of the buckets beyond the histogram bins.
void lowhash0BucketsAll(Context& ctx, const uint64_t* keysIn, const uint64_t* valsIn, uint64_t n,
    uint64_t* sendOffsets, const uint64_t** pairKeysOut, const uint32_t** pairTagsOut,
    uint64_t* bucketsUsedOut, uint64_t* sizeHistogramOut, std::vector<uint64_t>& overflowOut)
{
    LowHash0Job& job = jobOf(ctx);
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;
    const uint64_t I = job.p.minHashIterationCount;
    MI355X_ASSERT(I >= 1 && I <= 4096 && job.iterations == 0);
    if(n >= (1ULL << 32) - 1) throw std::runtime_error("LowHash0: this rank received " + std::to_string(n) + " low-hash records of all iterations, more than the 2^32 positions "
        "of one sort: run the job iteration after iteration (shasta_mi355x_lh_one_pass_fits says beforehand).");
    // SHASTA_MI355X_LOG_STAGES=1: a call of more than 25 ms says on stderr where it spent them (each mark after a stream
    // synchronisation of its own -- a diagnosis, not for timed runs).
    static const bool logStages = [] { const char* e = std::getenv("SHASTA_MI355X_LOG_STAGES"); return e && e[0] == '1'; }();
    std::vector<std::pair<const char*, double>> marks;
    const auto t0 = std::chrono::steady_clock::now();
    const auto mark = [&](const char* what) {
        if(!logStages) return;
        HIP_CHECK(hipStreamSynchronize(stream));
        marks.emplace_back(what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    mark("entered (stream idle)");
    reserveIterationRows(job, I, stream);
    mark("iteration rows cleared");
    job.recKeysA.reserve(2 * std::max<uint64_t>(n, 1), stream); job.recKeysB.reserve(2 * std::max<uint64_t>(n, 1), stream);
    job.recValsA.reserve(std::max<uint64_t>(n, 1), stream); job.recValsB.reserve(std::max<uint64_t>(n, 1), stream);
    if(n && (const void*)keysIn != (const void*)job.recKeysA.data()) HIP_CHECK(hipMemcpyAsync(job.recKeysA.data(), keysIn, n * 8, hipMemcpyDeviceToDevice, stream));
    mark("keys copied in");
    if(n && valsIn != job.recValsA.data()) HIP_CHECK(hipMemcpyAsync(job.recValsA.data(), valsIn, n * 8, hipMemcpyDeviceToDevice, stream));
    const uint32_t* keys = nullptr; const uint64_t* vals = nullptr;
    mark("records copied in");
    enqueueSortRecords(ctx, job, keys, vals, Count(n), uint32_t(I), true);
    mark("records sorted");
    unsigned long long* counters = job.counters.data();
    // The pair keys go where the single-GPU job keeps them; a guess that was too small: everything this call added is
    // zeroed and the buckets run again with room (the records stay sorted where they are).
    uint64_t pairCount = 0;
    for(int attempt = 0; ; attempt++) {
        MI355X_ASSERT(attempt < 4);
        HIP_CHECK(hipMemsetAsync(counters + C_PAIRS, 0, sizeof(unsigned long long), stream));
        HIP_CHECK(hipMemsetAsync(counters + C_BUCKETS, 0, sizeof(unsigned long long), stream));
        HIP_CHECK(hipMemsetAsync(counters + C_OVERFLOW, 0, sizeof(unsigned long long), stream));
        HIP_CHECK(hipMemsetAsync(job.stats.data(), 0, 3 * ctx.readCount * sizeof(unsigned long long), stream));
        HIP_CHECK(hipMemsetAsync(job.sizeHist.data(), 0, I * SIZE_HIST_CAP * sizeof(unsigned long long), stream));
        enqueueBuckets(ctx, job, keys, vals, Count(n), 0, job.pairKeys(), job.pairTags(), job.pairCapacity, uint32_t(I), true);
        pairCount = readDevice(counters + C_PAIRS, stream);
        mark("buckets, statistics, pair keys");
        if(pairCount <= job.pairCapacity) break;
        reservePairs(job, pairCount + pairCount / 8 + 64, stream, false);
        ctx.lowhashPairsHint = job.pairCapacity;
    }
    std::vector<unsigned long long> table(4 * I);
    HIP_CHECK(hipMemcpyAsync(table.data(), job.iterationTable.data(), table.size() * 8, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(sizeHistogramOut, job.sizeHist.data(), I * SIZE_HIST_CAP * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    const uint64_t overflow = readDevice(counters + C_OVERFLOW, stream);
    if(overflow > LowHash0Job::overflowCapacity) throw std::runtime_error("LowHash0: bucket-size overflow list exhausted.");
    overflowOut.assign(overflow, 0);
    if(overflow) HIP_CHECK(hipMemcpyAsync(overflowOut.data(), job.overflowSizes.data(), overflow * 8, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    mark("tables copied out");
    for(uint64_t t = 0; t < I; t++) bucketsUsedOut[t] = table[4 * t + 1];
    // Sorted by key with the tags: an owner's keys are contiguous (readId0 leads the key).
    const uint64_t* pk = job.pairKeys(); const uint32_t* tags = job.pairTags();
    if(pairCount) {
        MI355X_ASSERT(pairCount < (1ULL << 32) - 1);
        const KernelTimers::Span span = ctx.timers.begin("radix sort of the pair keys of all iterations (staged)", stream);
        const bool flipped = radixSort<uint64_t, uint32_t, true>(job.pairKeys(), job.pairsInB ? job.pairKeysA.data() : job.pairKeysB.data(),
            job.pairTags(), job.pairsInB ? job.pairTagsA.data() : job.pairTagsB.data(), pairCount, job.pairKeyBits, ctx.sortWs, stream);
        if(flipped) job.pairsInB = !job.pairsInB;
        pk = job.pairKeys(); tags = job.pairTags();
        (void)ctx.timers.end(span, 2 * 12 * pairCount * uint64_t((job.pairKeyBits + 7) / 8), pairCount);
    }
    if(job.world == 1 || pairCount == 0) {
        for(int r = 0; r <= job.world; r++) sendOffsets[r] = (r == job.world) ? pairCount : 0;
    } else {
        hipLaunchKernelGGL(lowerBoundsKernel<uint64_t>, dim3(divUp(uint64_t(job.world) + 1, 64)), dim3(64), 0, stream,
            pk, pairCount, (const uint64_t*)job.boundKeys64.data(), uint32_t(job.world + 1), job.boundOut.data());
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(sendOffsets, job.boundOut.data(), (size_t(job.world) + 1) * 8, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        sendOffsets[0] = 0; sendOffsets[job.world] = pairCount;
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    mark("pair keys sorted and split");
    if(logStages && !marks.empty() && marks.back().second > 25.) {
        std::string line = "shasta_mi355x: lh_buckets_all took " + std::to_string(marks.back().second) + " ms:";
        for(const auto& m : marks) line += std::string(" [") + m.first + " " + std::to_string(m.second) + "]";
        std::fprintf(stderr, "%s\n", line.c_str());
    }
    *pairKeysOut = pk; *pairTagsOut = tags;
}

// Stage 3 (all iterations).  The n pair keys whose readId0 this rank owns, with their iteration tags (device pointers: the
// exchange's receive buffers, or -- one rank -- what stage 2 returned).
void lowhash0MergeAll(Context& ctx, const uint64_t* keys, const uint32_t* tags, uint64_t n)
{
    LowHash0Job& job = jobOf(ctx);
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;
    MI355X_ASSERT(job.iterations == 0);
    if(n) {
        const bool own = keys == job.pairKeys() && tags == job.pairTags();
        if(!own) {
            if(n > job.pairCapacity) reservePairs(job, n + n / 8 + 64, stream, false);
            HIP_CHECK(hipMemcpyAsync(job.pairKeys(), keys, n * 8, hipMemcpyDeviceToDevice, stream));
            HIP_CHECK(hipMemcpyAsync(job.pairTags(), tags, n * 4, hipMemcpyDeviceToDevice, stream));
        }
    }
    job.pairCount = n;
    job.iterations = job.p.minHashIterationCount;
    // The evaluation replays a pair's history from the order of its occurrences after a STABLE sort by key: they have to be
    // in iteration order before it.  What arrived is one run per sender, each sorted by key: sort by iteration (stable).
    if(n && job.world > 1) {
        const bool flipped = radixSort<uint32_t, uint64_t, true>(job.pairTags(), job.pairsInB ? job.pairTagsA.data() : job.pairTagsB.data(),
            job.pairKeys(), job.pairsInB ? job.pairKeysA.data() : job.pairKeysB.data(), n, bitsFor(job.iterations - 1), ctx.sortWs, stream);
        if(flipped) job.pairsInB = !job.pairsInB;
    }
    HIP_CHECK(hipStreamSynchronize(stream));              // `keys` / `tags` may be reused by the caller
}

// Iterations merged so far by the job in progress (what lowhash0Finish will report per iteration).
uint64_t lowhash0JobIterations(Context& ctx) { return jobOf(ctx).iterations; }
uint64_t lowhash0JobPlannedIterations(Context& ctx) { return jobOf(ctx).p.minHashIterationCount; }

// Stage 4.  Candidates of this rank's readId0 range (sorted), statistics (this rank's partial sums, readCount x 3),
// this rank's share of the per-iteration counters; ends the job.
void lowhash0Finish(Context& ctx, uint64_t* readLowHashStatistics, std::vector<shasta_oriented_read_pair>& hostCandidates,
    std::vector<uint64_t>& highPerIteration, std::vector<uint64_t>& totalPerIteration)
{
    lowhash0Finish(ctx, readLowHashStatistics, &hostCandidates, highPerIteration, totalPerIteration, nullptr, nullptr);
}

// hostCandidates == nullptr: the candidates stay where the last kernel wrote them; *deviceCandidates / *deviceCandidateCount
// name them (valid until the context's next LowHash0 job begins: the retired job's allocations are kept for it).
void lowhash0Finish(Context& ctx, uint64_t* readLowHashStatistics, std::vector<shasta_oriented_read_pair>* hostCandidatesOrNull,
    std::vector<uint64_t>& highPerIteration, std::vector<uint64_t>& totalPerIteration,
    const shasta_oriented_read_pair** deviceCandidates, uint64_t* deviceCandidateCount)
{
    std::vector<shasta_oriented_read_pair> unused;
    std::vector<shasta_oriented_read_pair>& hostCandidates = hostCandidatesOrNull ? *hostCandidatesOrNull : unused;
    LowHash0Job& job = jobOf(ctx);
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;
    const uint64_t readCount = ctx.readCount;
    evaluate(ctx, job);
    highPerIteration = job.highHost; totalPerIteration = job.totalHost;
    hostCandidates.clear();
    // K6.
    if(job.candidateCount) {
        job.candidatesDevice.reserve(job.candidateCount, stream);
        SHASTA_TIMED(ctx, "emitCandidatesKernel", stream, 12 * job.candidateCount, job.candidateCount,
            hipLaunchKernelGGL(emitCandidatesKernel, dim3(divUp(job.pairCount, 256)), dim3(256), 0, stream,
                (const uint64_t*)job.pairKeys(), (const uint32_t*)job.pos.data(), job.pairCount, job.readBits, job.candidatesDevice.data()));
        HIP_CHECK(hipGetLastError());
        if(hostCandidatesOrNull) {
            hostCandidates.resize(job.candidateCount);
            HIP_CHECK(hipMemcpyAsync(hostCandidates.data(), job.candidatesDevice.data(),
                job.candidateCount * sizeof(shasta_oriented_read_pair), hipMemcpyDeviceToHost, stream));
        }
    }
    if(deviceCandidates) *deviceCandidates = job.candidateCount ? job.candidatesDevice.data() : nullptr;
    if(deviceCandidateCount) *deviceCandidateCount = job.candidateCount;
    HIP_CHECK(hipMemcpyAsync(readLowHashStatistics, job.stats.data(), 3 * readCount * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    ctx.lowhashPairsHint = std::max(ctx.lowhashPairsHint, job.pairCount + job.pairCount / 8);
    retireJob(ctx);
}

// The whole of LowHash0::LowHash0 on one GPU.
void lowhash0Run(Context& ctx, const shasta_lowhash0_params& p, uint64_t* readLowHashStatistics, shasta_lowhash0_result& result)
{
    std::memset(&result, 0, sizeof(result));
    const auto t0 = std::chrono::steady_clock::now();
    HIP_CHECK(hipSetDevice(ctx.device));
    hipStream_t stream = ctx.stream;
    const uint64_t readCount = ctx.readCount;
    const ScopedEvent evBegin, evEnd;
    HIP_CHECK(hipEventRecord(evBegin, stream));

    std::vector<uint64_t> highFrequencyPerIteration, totalPerIteration, histogramRows;
    std::vector<shasta_oriented_read_pair> hostCandidates;
    uint32_t log2BucketCount = 0;
    try {
        for(int attempt = 0; ; attempt++) {
            MI355X_ASSERT(attempt < 8);
            lowhash0Begin(ctx, p, 0, 1, nullptr, &log2BucketCount);
            LowHash0Job& job = jobOf(ctx);
            unsigned long long* counters = job.counters.data();
            unsigned long long host[C_COUNT];
            bool again = false;
            uint64_t highFrequency = 0;
            // All iterations in one pass over the markers (hashWindowsKernel<m, true>) whenever their number is known in
            // advance (with the dynamic iteration control: iteration after iteration as before); the records' sort key is
            // iteration | bucket id in 32 bits where that fits, a 64-bit key beyond (2^31 buckets, the human-genome value).
            // SHASTA_MI355X_LOWHASH_ONE_PASS=0: never.
            const bool onePassAllowed = [] { const char* e = std::getenv("SHASTA_MI355X_LOWHASH_ONE_PASS"); return !(e && e[0] == '0'); }();      // (read for every call: tests switch it)
            const uint64_t I = p.minHashIterationCount;
            const bool onePass = onePassAllowed && I >= 1 && I <= 4096 && I * job.recCapacity < onePassRecordLimit();
            if(onePassAllowed && !onePass && I >= 1) noteOnePassFallback(I, job.recCapacity);
            const bool wideKeys = onePass && job.log2BucketCount + uint64_t(bitsFor(I - 1)) > 32;      // 64-bit record keys: iteration << 32 | bucket id
            uint64_t recordCapacityAll = 0;
            if(onePass) {
                reserveIterationRows(job, I, stream);
                recordCapacityAll = I * job.recCapacity;
                const uint64_t keyWords = recordCapacityAll * (wideKeys ? 2 : 1);
                job.recKeysA.reserve(keyWords, stream); job.recKeysB.reserve(keyWords, stream);
                job.recValsA.reserve(recordCapacityAll, stream); job.recValsB.reserve(recordCapacityAll, stream);
                const KernelTimers::Span span = ctx.timers.begin(hashKernelName(uint32_t(p.m), true, wideKeys), stream);
                if(wideKeys) launchHash<uint64_t>(ctx, uint32_t(p.m), 0, job.hashThreshold, job.mask, job.markerBegin, job.markerEnd,
                    reinterpret_cast<uint64_t*>(job.recKeysA.data()), job.recValsA.data(), counters + C_RECORDS, recordCapacityAll, uint32_t(I), 32u);
                else launchHash(ctx, uint32_t(p.m), 0, job.hashThreshold, job.mask, job.markerBegin, job.markerEnd,
                    job.recKeysA.data(), job.recValsA.data(), counters + C_RECORDS, recordCapacityAll, uint32_t(I), uint32_t(job.log2BucketCount));
                job.hashHandles.push_back(ctx.timers.end(span, 4 * (job.markerEnd - job.markerBegin), (job.markerEnd - job.markerBegin) * I));
                const uint32_t* keys = nullptr; const uint64_t* vals = nullptr;
                const Count records(recordCapacityAll, counters + C_RECORDS);
                enqueueSortRecords(ctx, job, keys, vals, records, uint32_t(I), wideKeys);
                enqueueBuckets(ctx, job, keys, vals, records, 0, job.pairKeys(), job.pairTags(), job.pairCapacity, uint32_t(I), wideKeys);
                job.iterations = I;
            }
            else for(uint64_t iteration = 0; ; iteration++) {
                // Iteration control, src/LowHash0.cpp:136-157.
                if(p.minHashIterationCount == 0) {
                    const double current = 2. * double(highFrequency) / double(readCount);
                    if(current >= p.alignmentCandidatesPerRead) break;
                } else if(iteration == p.minHashIterationCount) {
                    break;
                }
                reserveIterationRows(job, iteration + 1, stream);
                enqueueHash(ctx, job, iteration);
                const uint32_t* keys = nullptr; const uint64_t* vals = nullptr;
                const Count records(job.recCapacity, counters + C_RECORDS);
                enqueueSortRecords(ctx, job, keys, vals, records);
                enqueueBuckets(ctx, job, keys, vals, records, iteration, job.pairKeys(), job.pairTags(), job.pairCapacity);
                job.iterations = iteration + 1;
                if(p.minHashIterationCount == 0) {
                    // The iteration control needs this iteration's high-frequency count: read back, evaluate.
                    HIP_CHECK(hipMemcpyAsync(host, counters, sizeof(host), hipMemcpyDeviceToHost, stream));
                    HIP_CHECK(hipStreamSynchronize(stream));
                    if(host[C_MAX_RECORDS] > job.recCapacity || host[C_PAIRS] > job.pairCapacity) { again = true; break; }
                    job.pairCount = host[C_PAIRS];
                    evaluate(ctx, job);
                    highFrequency = job.highHost.back();
                    // Room for the next iteration's keys, judged by this one's (checked again afterwards).
                    const uint64_t perIteration = job.pairCount / job.iterations + 1;
                    if(job.pairCount + 2 * perIteration > job.pairCapacity) reservePairs(job, 2 * (job.pairCount + 2 * perIteration), stream, true);
                }
            }
            HIP_CHECK(hipMemcpyAsync(host, counters, sizeof(host), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            if(onePass) host[C_MAX_RECORDS] = host[C_TOTAL_RECORDS] > recordCapacityAll ? (host[C_TOTAL_RECORDS] + I - 1) / I + 1 : 0;      // (per iteration, for the hint)
            if(host[C_MAX_RECORDS] > job.recCapacity || host[C_PAIRS] > job.pairCapacity) again = true;
            if(again) {
                // A capacity guess was too small (keys or records were dropped, counters are exact): once more with room.
                ctx.lowhashRecordsHint = std::max<uint64_t>(ctx.lowhashRecordsHint, host[C_MAX_RECORDS] + host[C_MAX_RECORDS] / 4);
                const uint64_t iterationsDone = std::max<uint64_t>(1, job.iterations);
                const uint64_t planned = p.minHashIterationCount ? p.minHashIterationCount : 2 * iterationsDone;
                ctx.lowhashPairsHint = std::max<uint64_t>(ctx.lowhashPairsHint, (host[C_PAIRS] / iterationsDone + 1) * planned * 5 / 4);
                retireJob(ctx);
                continue;
            }
            if(host[C_OVERFLOW] > LowHash0Job::overflowCapacity) throw std::runtime_error("LowHash0: bucket-size overflow list exhausted.");
            job.pairCount = host[C_PAIRS];
            const uint64_t iterations = job.iterations;
            // Per-iteration counters, histograms, the list of bucket sizes beyond the histogram bins.
            std::vector<unsigned long long> table(4 * std::max<uint64_t>(1, iterations)), hist(SIZE_HIST_CAP * std::max<uint64_t>(1, iterations)), overflowEntries(host[C_OVERFLOW]);
            if(iterations) {
                HIP_CHECK(hipMemcpyAsync(table.data(), job.iterationTable.data(), 4 * iterations * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipMemcpyAsync(hist.data(), job.sizeHist.data(), SIZE_HIST_CAP * iterations * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            }
            if(!overflowEntries.empty()) HIP_CHECK(hipMemcpyAsync(overflowEntries.data(), job.overflowSizes.data(), overflowEntries.size() * 8, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            const uint64_t bucketCount = 1ULL << log2BucketCount;
            for(uint64_t iteration = 0; iteration < iterations; iteration++) {
                std::vector<uint32_t> overflow;
                for(unsigned long long e : overflowEntries) if((e >> 32) == iteration) overflow.push_back(uint32_t(e));
                static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "u64");
                appendHistogramRows(iteration, bucketCount, table[4 * iteration + 1], reinterpret_cast<const uint64_t*>(hist.data()) + iteration * SIZE_HIST_CAP, overflow, histogramRows);
                // Algorithmic bytes of the iteration's hash launch: 4 B per marker read + 12 B per low hash written.
                if(!onePass && iteration < job.hashHandles.size()) ctx.timers.amend(job.hashHandles[iteration], 4 * (job.markerEnd - job.markerBegin) + 12 * table[4 * iteration], job.markerEnd - job.markerBegin);
            }
            // (one pass: the markers are read once, the low hashes of every iteration written)
            if(onePass && !job.hashHandles.empty()) ctx.timers.amend(job.hashHandles[0], 4 * (job.markerEnd - job.markerBegin) + 12 * host[C_TOTAL_RECORDS], (job.markerEnd - job.markerBegin) * I);
            lowhash0Finish(ctx, readLowHashStatistics, hostCandidates, highFrequencyPerIteration, totalPerIteration);
            break;
        }
    } catch(...) {
        ctx.lowhashJob.reset();
        throw;
    }
    HIP_CHECK(hipEventRecord(evEnd, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, evBegin, evEnd));
    result.deviceSeconds = ms * 1e-3;

    result.log2BucketCount = log2BucketCount;
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
