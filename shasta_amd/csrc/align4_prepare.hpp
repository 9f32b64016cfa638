// Align4 on MI355X: which table class a candidate's cells run in (one function for the host and the device), and the
// preparation of a batch's first round on the device -- classes, the grouping of the candidates by (class, tabled oriented
// read) and the chunk lists of the three classes -- where the host did it with a loop, a sort and a walk over the sorted
// list before the batch's first kernel could start.  Included by align4.hip inside its anonymous namespace, after the class
// geometry (CELLS_NA_LOG2, CELLS_SC_LOG2, CELLS_CHUNK_MAX).
#pragma once

// (constexpr arrays cannot be indexed by a run-time value in device code; these can)
__host__ __device__ inline int cellsNaLog2(int c) { return c == 0 ? CELLS_NA_LOG2[0] : (c == 1 ? CELLS_NA_LOG2[1] : (c == 2 ? CELLS_NA_LOG2[2] : (c == 3 ? CELLS_NA_LOG2[3] : CELLS_NA_LOG2[4]))); }
__host__ __device__ inline int cellsScLog2(int c) { return c == 0 ? CELLS_SC_LOG2[0] : (c == 1 ? CELLS_SC_LOG2[1] : (c == 2 ? CELLS_SC_LOG2[2] : (c == 3 ? CELLS_SC_LOG2[3] : CELLS_SC_LOG2[4]))); }
__host__ __device__ inline uint32_t cellsChunkMax(int c) { return c == 0 ? CELLS_CHUNK_MAX[0] : (c == 1 ? CELLS_CHUNK_MAX[1] : (c == 2 ? CELLS_CHUNK_MAX[2] : (c == 3 ? CELLS_CHUNK_MAX[3] : (c == 4 ? CELLS_CHUNK_MAX[4] : CELLS_CHUNK_MAX[5])))); }
static_assert(CELLS_CLASSES == 6 && CELLS_LONG == 4 && CELLS_LONG_BIG == 5 && CELLS_NA_LOG2[4] == CELLS_NA_LOG2[5] && CELLS_SC_LOG2[4] == CELLS_SC_LOG2[5],
    "cellsNaLog2 / cellsScLog2 / cellsChunkMax name the five classes a candidate starts in, the last one the windowed one (its large-graph form is never a first choice)");
constexpr int CELLS_CLASS_BITS = 3;                         // the class (0 .. CELLS_CLASSES, the last = HBM scratch) in the sort key

// estimateShift: a candidate's random background is about nx ny >> estimateShift matches (Context::matchShift: from a sample of the
// read set's markers; 13 for the k = 10 alphabets, 18 at k = 14).
// force: 0, or (SHASTA_MI355X_CELLS_FORCE=long / big: the tests' switch) 1 / 2 = every candidate the windowed class can take starts in it / in its
// large-graph form, whatever smaller class would do -- the whole suite's read sets through those two kernels.
struct CellsClassRule { uint64_t deltaX, deltaY; bool packedOk, longOk; int estimateShift; int force; };
inline CellsClassRule cellsClassRule(const DeviceOptions& opt, int estimateShift)
{
    CellsClassRule r;
    r.deltaX = opt.deltaX; r.deltaY = opt.deltaY;
    // The packed LDS cell word counts up to 2^CELLS_COUNT_BITS - 1 entries; a cell holds at most
    // ceil(deltaX * deltaY / 2) (one (x,y) per lattice point of the right parity).
    r.packedOk = (uint64_t(opt.deltaX) * opt.deltaY + 1) / 2 < (1ULL << CELLS_COUNT_BITS) && opt.deltaX >= 2 && opt.deltaY >= 2;
    // The windowed class counts in 8 bits and stops adding at the threshold (align4_cells.hpp): thresholds up to 191 leave the
    // adds in flight room (a count that reaches 255 all the same is detected and the candidate climbs to the HBM-scratch kernel).
    r.longOk = opt.deltaX >= 2 && opt.deltaY >= 2 && opt.minEntryCountPerCell <= 191;
    r.estimateShift = estimateShift;
    r.force = 0;
    if(const char* e = std::getenv("SHASTA_MI355X_CELLS_FORCE")) r.force = std::string(e) == "long" ? 1 : (std::string(e) == "big" ? 2 : 0);      // (read for every batch: tests switch it)
    return r;
}
// Class of a candidate that tables a read of `tabled` markers: table of the tabled read at load <= 1/2, cell table sized for
// the expected number of distinct cells (random background ~ nx*ny / alphabet, plus the diagonal) at load <= 3/4.  Overflow is
// detected on the device and climbs one class.  CELLS_LONG: the class that tables the shorter read in windows of 2^13 markers
// (either read of any length below 65 535; wider cell indices).  CELLS_CLASSES: the kernel with its tables in HBM scratch.
__host__ __device__ inline int cellsClassFor(const CellsClassRule& rule, uint64_t tabled, uint64_t nx, uint64_t ny)
{
    if(nx >= 65535 || ny >= 65535) return CELLS_CLASSES;
    // The single-multiply division must be exact.
    if((nx + ny) * (rule.deltaX > rule.deltaY ? rule.deltaX : rule.deltaY) >= (1ULL << 32)) return CELLS_CLASSES;
    const uint64_t cells = (nx * ny >> rule.estimateShift) + (nx + ny) / 32 + 32;
    // (the windowed class's own estimate is the expected number of cells, not twice it: what overflows its table goes on to the HBM-scratch kernel
    // either way, and a pair of two reads of 40 000 markers -- 17 a step of the ultra-long shape -- has 2 600 random matches at k = 14, not 6 100)
    const uint64_t cellsLong = (nx * ny >> (rule.estimateShift + 1)) + (nx + ny) / 64 + 32;
    const bool longFits = rule.longOk && (nx + ny) / rule.deltaX < (1ULL << CELLS_IX_BITS) && (nx + ny) / rule.deltaY < (1ULL << CELLS_LONG_IY_BITS) - 1 && 4 * cellsLong <= (3ULL << cellsScLog2(CELLS_LONG));
    if(rule.force && longFits) return rule.force == 2 ? CELLS_LONG_BIG : CELLS_LONG;
    // Cell indices must fit the packed word.
    if(rule.packedOk && (nx + ny) / rule.deltaX < (1ULL << CELLS_IX_BITS) && (nx + ny) / rule.deltaY < (1ULL << CELLS_IY_BITS)) {
        for(int c = 0; c < CELLS_LONG; c++) {
            if(tabled < (1ULL << cellsNaLog2(c)) && 4 * cells <= (3ULL << cellsScLog2(c))) return c;
        }
    }
    if(longFits) return CELLS_LONG;
    return CELLS_CLASSES;
}
// What the windowed class asks of a candidate's geometry (not the estimate of its cells): for a candidate that climbs to its large-graph form.
__host__ __device__ inline bool cellsLongGeometryOk(const CellsClassRule& rule, uint64_t nx, uint64_t ny)
{
    return nx < 65535 && ny < 65535 && rule.longOk && (nx + ny) * (rule.deltaX > rule.deltaY ? rule.deltaX : rule.deltaY) < (1ULL << 32) &&
        (nx + ny) / rule.deltaX < (1ULL << CELLS_IX_BITS) && (nx + ny) / rule.deltaY < (1ULL << CELLS_LONG_IY_BITS) - 1;
}
// Every candidate tables whichever of its two reads lands in the smaller class (ties: read 0).  (The windowed class tables the
// shorter read whatever this says: align4CellsLongKernel decides for itself.)
struct CellsChoice { int cls; bool swapped; };
__host__ __device__ inline CellsChoice cellsChoice(const CellsClassRule& rule, uint32_t nx, uint32_t ny)
{
    const int c0 = cellsClassFor(rule, nx, nx, ny);
    const int c1 = ny < nx ? cellsClassFor(rule, ny, nx, ny) : CELLS_CLASSES;
    CellsChoice r;
    r.swapped = c1 < c0;
    r.cls = r.swapped ? c1 : c0;
    if(r.cls == CELLS_LONG || r.cls == CELLS_LONG_BIG) r.swapped = ny < nx;
    return r;
}

// ---- the first round's lists on the device ------------------------------------------------------------------------------
// Sort key of a candidate: class | swapped | the tabled oriented read's id (tabledBits bits).  A STABLE sort leaves
// the candidates of a (class, swapped, tabled read) group adjacent and in ascending order -- the groups the host walk made,
// class by class -- and the candidates of the HBM-scratch kernel (class CELLS_CLASSES) at the end.
// info: [0 .. CELLS_CLASSES] = first chunk of every class and the number of chunks; [CELLS_INFO_FIRST_BIG] = position of the first
//       HBM-scratch candidate in the sorted list; [CELLS_INFO_CANDIDATES + c] = candidates of class c; [CELLS_INFO_BYTES + c] =
//       their algorithmic bytes, 4 (nx + ny) each.
constexpr int CELLS_INFO_FIRST_BIG = CELLS_CLASSES + 1, CELLS_INFO_CANDIDATES = CELLS_CLASSES + 2, CELLS_INFO_BYTES = 2 * CELLS_CLASSES + 2;
constexpr int CELLS_PREPARE_INFO = 3 * CELLS_CLASSES + 2;

__global__ void __launch_bounds__(256)
cellsClassKeysKernel(const PairDesc* __restrict__ pairs, const shasta_oriented_read_pair* __restrict__ candidates, uint32_t n, CellsClassRule rule, int tabledBits,
    uint64_t* __restrict__ keys, uint32_t* __restrict__ ids, unsigned long long* __restrict__ info)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1;
    unsigned long long bytes = 0;
    if(q < n) {
        const PairDesc pd = pairs[q];
        const CellsChoice choice = cellsChoice(rule, pd.nx, pd.ny);
        cls = choice.cls;
        // The tabled oriented read by its ID (the same order as by its first marker, in 18 bits where that takes 29: three passes
        // of the sort below instead of five).
        const shasta_oriented_read_pair c = candidates[q];
        // (the windowed class tables every candidate's read anew: its chunks are any sixteen candidates, one group)
        const bool grouped = cls != CELLS_LONG && cls != CELLS_LONG_BIG;
        const uint64_t tabled = !grouped ? 0ULL : (choice.swapped ? (2ULL * c.readIds[1] + (c.isSameStrand ? 0u : 1u)) : 2ULL * c.readIds[0]);
        keys[q] = (uint64_t(cls) << (tabledBits + 1)) | (uint64_t(grouped && choice.swapped ? 1 : 0) << tabledBits) | tabled;
        ids[q] = q;
        bytes = 4ULL * (uint64_t(pd.nx) + pd.ny);
    }
#pragma unroll
    for(int c = 0; c < CELLS_CLASSES; c++) {
        const uint64_t votes = __ballot(cls == c);
        if(votes == 0) continue;
        unsigned long long classBytes = cls == c ? bytes : 0;
        for(int d = 32; d >= 1; d >>= 1) classBytes += __shfl_down(classBytes, d, WAVE);
        if(laneId() == 0) {
            atomicAdd(&info[CELLS_INFO_CANDIDATES + c], (unsigned long long)__popcll(votes));
            atomicAdd(&info[CELLS_INFO_BYTES + c], classBytes);
        }
    }
}

// First position of the sorted keys that is not below `key`.
__device__ inline uint32_t cellsLowerBound(const uint64_t* __restrict__ keys, uint32_t n, uint64_t key)
{
    uint32_t lo = 0, hi = n;
    while(lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if(keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// flags[i] = 1 where a chunk begins: every cellsChunkMax(class) candidates of a group; flags[n] = 0.
__global__ void __launch_bounds__(256)
cellsChunkHeadsKernel(const uint64_t* __restrict__ sortedKeys, uint32_t n, int tabledBits, uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i > n) return;
    uint32_t head = 0;
    if(i < n) {
        const uint64_t key = sortedKeys[i];
        const int cls = int(key >> (tabledBits + 1));
        if(cls < CELLS_CLASSES) head = ((i - cellsLowerBound(sortedKeys, n, key)) % cellsChunkMax(cls)) == 0 ? 1u : 0u;
    }
    flags[i] = head;
}

// ranks = exclusive scan of the flags over n + 1 elements.
__global__ void __launch_bounds__(256)
cellsChunkWriteKernel(const uint64_t* __restrict__ sortedKeys, const uint32_t* __restrict__ ranks, uint32_t n, int tabledBits,
    CellsChunk* __restrict__ chunks, unsigned long long* __restrict__ info)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n && ranks[i + 1] != ranks[i]) {
        const uint64_t key = sortedKeys[i];
        const int cls = int(key >> (tabledBits + 1));
        const uint32_t groupEnd = cellsLowerBound(sortedKeys, n, key + 1);
        CellsChunk ch;
        ch.firstMember = i; ch.count = uint16_t(min(cellsChunkMax(cls), groupEnd - i)); ch.swapped = uint16_t((key >> tabledBits) & 1);
        ch.naLog2 = uint32_t(cellsNaLog2(cls)); ch.scLog2 = uint32_t(cellsScLog2(cls));
        chunks[ranks[i]] = ch;
    }
    if(i <= uint32_t(CELLS_CLASSES)) {
        const uint32_t first = cellsLowerBound(sortedKeys, n, uint64_t(i) << (tabledBits + 1));      // first candidate of class i
        info[i] = ranks[first];
        if(i == uint32_t(CELLS_CLASSES)) info[CELLS_INFO_FIRST_BIG] = first;
    }
}
