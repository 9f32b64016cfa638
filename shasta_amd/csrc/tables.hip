// SURVEY 8(f) row 3, the serial tails either side of the aligner, on the device:
//   pairTable      the table "oriented read -> indices of the pairs it takes part in, sorted by (other oriented read,
//                  index)" that Assembler::computeAlignmentTable (src/AssemblerAlign.cpp:509-571) builds from AlignmentData
//                  and AlignmentCandidates::computeCandidateTable (src/AssemblerAlignmentCandidates.cpp:388-447) from the
//                  candidate list: two counting passes and a std::sort per oriented read there, ONE stable radix sort of
//                  the 4 N (oriented read, other oriented read) keys here;
//   readGraphKeep  createReadGraph's selection (src/AssemblerReadGraph.cpp:55-95): alignment i stays if it is among the
//                  maxAlignmentCount best of either of its reads, best = largest (markerCount, alignment id): an
//                  nth_element per read there, two stable radix sorts (by (markerCount, id) descending, then by read) and a
//                  rank here.
// Both are HBM-bound sorts of a few million 12-byte records: nothing to tune beyond primitives.hpp's radix sort.
#include "context.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

namespace shasta_mi355x {
namespace {

// The pair at byte offset i * stride of `pairs` (12-byte candidates, or the head of 64-byte AlignmentData rows).
__device__ __forceinline__ shasta_oriented_read_pair pairAt(const uint8_t* __restrict__ pairs, uint64_t stride, uint64_t i)
{
    const uint32_t* p = reinterpret_cast<const uint32_t*>(pairs + i * stride);
    shasta_oriented_read_pair r;
    r.readIds[0] = p[0]; r.readIds[1] = p[1];
    r.isSameStrand = reinterpret_cast<const uint8_t*>(p + 2)[0];
    return r;
}

// The four (oriented read, partner) entries of pair i (OrientedReadPair::getOther, src/OrientedReadPair.hpp:63-85:
// the partner is reverse complemented where the read is).
__global__ void __launch_bounds__(256)
pairTableKeysKernel(const uint8_t* __restrict__ pairs, uint64_t stride, uint64_t count, uint64_t orientedReadCount, int otherBits,
    uint64_t* __restrict__ keys, uint32_t* __restrict__ values, uint32_t* __restrict__ bad)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i >= count) return;
    const shasta_oriented_read_pair pair = pairAt(pairs, stride, i);
    const uint64_t o0 = uint64_t(pair.readIds[0]) << 1, o1 = (uint64_t(pair.readIds[1]) << 1) | (pair.isSameStrand ? 0u : 1u);
    if(o0 >= orientedReadCount || o1 >= orientedReadCount) { atomicAdd(bad, 1u); }
    const uint64_t rows[4] = {o0, o1, o0 ^ 1u, o1 ^ 1u}, others[4] = {o1, o0, o1 ^ 1u, o0 ^ 1u};
#pragma unroll
    for(int k = 0; k < 4; k++) { keys[4 * i + k] = (rows[k] << otherBits) | others[k]; values[4 * i + k] = uint32_t(i); }
}

// toc[r] = number of sorted keys whose oriented read is below r (r = 0 .. orientedReadCount).
__global__ void __launch_bounds__(256)
rowStartsKernel(const uint64_t* __restrict__ sortedKeys, uint64_t keyCount, uint64_t rowCount, int shift, uint64_t* __restrict__ toc)
{
    const uint64_t r = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(r > rowCount) return;
    uint64_t lo = 0, hi = keyCount;
    while(lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if((sortedKeys[mid] >> shift) < r) lo = mid + 1; else hi = mid;
    }
    toc[r] = lo;
}

// ---- the same table for more keys than one radix sort takes (4 N >= 2^32, or the test switch): the oriented reads are cut
// into ranges of at most `keyLimit` entries, the pairs come through the device in slabs, a range's entries are laid out in
// pair order (counts per pair, one scan) and sorted on their own; rows are ascending across ranges, so the sorted ranges
// one after the other ARE the table.

__device__ __forceinline__ void pairEntries(const shasta_oriented_read_pair& pair, uint64_t rows[4], uint64_t others[4])
{
    const uint64_t o0 = uint64_t(pair.readIds[0]) << 1, o1 = (uint64_t(pair.readIds[1]) << 1) | (pair.isSameStrand ? 0u : 1u);
    rows[0] = o0; rows[1] = o1; rows[2] = o0 ^ 1u; rows[3] = o1 ^ 1u;
    others[0] = o1; others[1] = o0; others[2] = o1 ^ 1u; others[3] = o0 ^ 1u;
}

// rowCounts[r] += the entries of oriented read r in this slab.
__global__ void __launch_bounds__(256)
pairTableRowCountsKernel(const uint8_t* __restrict__ pairs, uint64_t stride, uint64_t count, uint64_t orientedReadCount,
    unsigned long long* __restrict__ rowCounts, uint32_t* __restrict__ bad)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i >= count) return;
    uint64_t rows[4], others[4];
    pairEntries(pairAt(pairs, stride, i), rows, others);
    if(rows[0] >= orientedReadCount || rows[1] >= orientedReadCount) { atomicAdd(bad, 1u); return; }
#pragma unroll
    for(int k = 0; k < 4; k++) atomicAdd(rowCounts + rows[k], 1ULL);
}

// inRange[i] = how many of pair i's four entries belong to the oriented reads [rowBegin, rowEnd).
__global__ void __launch_bounds__(256)
pairTableInRangeKernel(const uint8_t* __restrict__ pairs, uint64_t stride, uint64_t count, uint64_t rowBegin, uint64_t rowEnd,
    uint32_t* __restrict__ inRange)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i >= count) return;
    uint64_t rows[4], others[4];
    pairEntries(pairAt(pairs, stride, i), rows, others);
    uint32_t c = 0;
#pragma unroll
    for(int k = 0; k < 4; k++) c += (rows[k] >= rowBegin && rows[k] < rowEnd) ? 1u : 0u;
    inRange[i] = c;
}

// The entries of the range, in pair order: entry = position[i] (the scan of inRange) past `cursor`, the range's entries of
// the slabs before this one.
__global__ void __launch_bounds__(256)
pairTableRangeKeysKernel(const uint8_t* __restrict__ pairs, uint64_t stride, uint64_t count, uint64_t firstPair, uint64_t rowBegin, uint64_t rowEnd,
    int otherBits, const uint32_t* __restrict__ position, uint64_t cursor, uint64_t* __restrict__ keys, uint32_t* __restrict__ values)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i >= count) return;
    uint64_t rows[4], others[4];
    pairEntries(pairAt(pairs, stride, i), rows, others);
    uint64_t at = cursor + position[i];
#pragma unroll
    for(int k = 0; k < 4; k++) {
        if(rows[k] >= rowBegin && rows[k] < rowEnd) {
            keys[at] = ((rows[k] - rowBegin) << otherBits) | others[k];
            values[at] = uint32_t(firstPair + i);
            ++at;
        }
    }
}

// Two entries per alignment: (read, quality) with quality = (markerCount, alignment id) complemented, so that ascending
// order is best first.
__global__ void __launch_bounds__(256)
readGraphKeysKernel(const shasta_alignment_data* __restrict__ alignmentData, uint64_t count, uint64_t readCount,
    uint64_t* __restrict__ quality, uint32_t* __restrict__ reads, uint32_t* __restrict__ ids, uint32_t* __restrict__ bad)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i >= count) return;
    const shasta_alignment_data a = alignmentData[i];
    if(a.pair.readIds[0] >= readCount || a.pair.readIds[1] >= readCount) atomicAdd(bad, 1u);
    const uint64_t q = ~((uint64_t(a.info.markerCount) << 32) | uint64_t(uint32_t(i)));
#pragma unroll
    for(int k = 0; k < 2; k++) { quality[2 * i + k] = q; reads[2 * i + k] = a.pair.readIds[k]; ids[2 * i + k] = uint32_t(i); }
}

__global__ void __launch_bounds__(256)
fillIotaKernel(uint32_t* __restrict__ out, uint64_t n)
{
    const uint64_t k = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(k < n) out[k] = uint32_t(k);
}

__global__ void __launch_bounds__(256)
gatherReadsKernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ reads, uint64_t n, uint32_t* __restrict__ out)
{
    const uint64_t k = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(k < n) out[k] = reads[order[k]];
}

// Entry k of the list sorted by (read, quality): its rank among its read's entries decides.
__global__ void __launch_bounds__(256)
readGraphKeepKernel(const uint32_t* __restrict__ sortedReads, const uint32_t* __restrict__ sortedEntries, const uint32_t* __restrict__ ids,
    uint64_t n, uint32_t maxAlignmentCount, uint8_t* __restrict__ keep)
{
    const uint64_t k = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(k >= n) return;
    const uint32_t read = sortedReads[k];
    // First entry of this read: binary search (the list is sorted by read).
    uint64_t lo = 0, hi = k;
    while(lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if(sortedReads[mid] < read) lo = mid + 1; else hi = mid;
    }
    if(k - lo < uint64_t(maxAlignmentCount)) keep[ids[sortedEntries[k]]] = 1;
}

int bitsFor(uint64_t values)            // bits that hold 0 .. values - 1
{
    int b = 1;
    while((1ULL << b) < values && b < 63) ++b;
    return b;
}

}  // namespace

namespace {

// The table of a pair list whose 4 N entries are more than one sort takes (see the kernels above).  keyLimit: entries per
// sorted range and (a quarter of it) pairs per slab.
void pairTableInRanges(const uint8_t* pairs, uint64_t stride, uint64_t count, uint64_t rows, int otherBits, uint64_t keyLimit,
    uint64_t* toc, uint32_t* values, hipStream_t stream)
{
    const uint64_t slabPairs = std::max<uint64_t>(1, std::min<uint64_t>(count, keyLimit / 4));
    const uint64_t slabCount = (count + slabPairs - 1) / slabPairs;
    DeviceBuffer<uint8_t> slab;
    DeviceBuffer<uint64_t> rowCounts, scanTemp64;
    DeviceBuffer<uint32_t> inRange, scanTemp32, bad, valuesA, valuesB;
    DeviceBuffer<uint64_t> keysA, keysB;
    RadixSortWorkspace ws;
    slab.reserve(slabPairs * stride, stream); rowCounts.reserve(rows + 1, stream); scanTemp64.reserve(scanTempElements(rows + 1), stream);
    inRange.reserve(slabPairs + 1, stream); scanTemp32.reserve(scanTempElements(slabPairs + 1), stream); bad.reserve(1, stream);
    uint64_t resident = ~0ULL;                         // the slab that is on the device (a list of one slab is uploaded once)
    const auto bring = [&](uint64_t k) -> uint64_t {   // -> pairs in slab k
        const uint64_t first = k * slabPairs, m = std::min(slabPairs, count - first);
        if(resident != k) { HIP_CHECK(hipMemcpyAsync(slab.data(), pairs + first * stride, m * stride, hipMemcpyHostToDevice, stream)); resident = k; }
        return m;
    };

    // Entries per oriented read, their prefix sums = the table of contents.
    HIP_CHECK(hipMemsetAsync(rowCounts.data(), 0, (rows + 1) * sizeof(uint64_t), stream));
    HIP_CHECK(hipMemsetAsync(bad.data(), 0, sizeof(uint32_t), stream));
    for(uint64_t k = 0; k < slabCount; k++) {
        const uint64_t m = bring(k);
        hipLaunchKernelGGL(pairTableRowCountsKernel, dim3(divUp(m, 256)), dim3(256), 0, stream, (const uint8_t*)slab.data(), stride, m, rows, reinterpret_cast<unsigned long long*>(rowCounts.data()), bad.data());
        HIP_CHECK(hipGetLastError());
        if(slabCount > 1) HIP_CHECK(hipStreamSynchronize(stream));      // (the next upload overwrites the slab)
    }
    exclusiveScan<uint64_t>(rowCounts.data(), rowCounts.data(), rows + 1, scanTemp64.data(), stream);
    uint32_t hostBad = 0;
    HIP_CHECK(hipMemcpyAsync(&hostBad, bad.data(), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(toc, rowCounts.data(), (rows + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if(hostBad) throw std::runtime_error("pair_table: a pair names a read beyond readCount.");
    MI355X_ASSERT(toc[rows] == 4 * count);

    // Ranges of oriented reads of at most keyLimit entries, each sorted on its own.
    uint64_t rowBegin = 0;
    while(rowBegin < rows) {
        uint64_t rowEnd = std::upper_bound(toc + rowBegin, toc + rows + 1, toc[rowBegin] + keyLimit) - toc - 1;       // the last r with toc[r] - toc[rowBegin] <= keyLimit
        if(rowEnd == rowBegin) throw std::runtime_error("pair_table: one oriented read takes part in more pairs than one sort takes.");
        const uint64_t base = toc[rowBegin], m = toc[rowEnd] - base;
        if(m) {
            keysA.reserve(m, stream); keysB.reserve(m, stream); valuesA.reserve(m, stream); valuesB.reserve(m, stream);
            uint64_t cursor = 0;
            for(uint64_t k = 0; k < slabCount; k++) {
                const uint64_t n = bring(k);
                hipLaunchKernelGGL(pairTableInRangeKernel, dim3(divUp(n, 256)), dim3(256), 0, stream, (const uint8_t*)slab.data(), stride, n, rowBegin, rowEnd, inRange.data());
                HIP_CHECK(hipMemsetAsync(inRange.data() + n, 0, sizeof(uint32_t), stream));
                exclusiveScan<uint32_t>(inRange.data(), inRange.data(), n + 1, scanTemp32.data(), stream);                // (element n = the slab's total)
                hipLaunchKernelGGL(pairTableRangeKeysKernel, dim3(divUp(n, 256)), dim3(256), 0, stream, (const uint8_t*)slab.data(), stride, n, k * slabPairs, rowBegin, rowEnd,
                    otherBits, (const uint32_t*)inRange.data(), cursor, keysA.data(), valuesA.data());
                HIP_CHECK(hipGetLastError());
                uint32_t total = 0;
                HIP_CHECK(hipMemcpyAsync(&total, inRange.data() + n, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
                cursor += total;
            }
            MI355X_ASSERT(cursor == m);
            const bool inB = radixSort<uint64_t, uint32_t, true>(keysA.data(), keysB.data(), valuesA.data(), valuesB.data(), m,
                otherBits + bitsFor(std::max<uint64_t>(rowEnd - rowBegin, 2)), ws, stream);
            HIP_CHECK(hipMemcpyAsync(values + base, inB ? valuesB.data() : valuesA.data(), m * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
        }
        rowBegin = rowEnd;
    }
}

}  // namespace

// toc: uint64[2 readCount + 1]; values: uint32[4 count].
void pairTable(int device, const void* pairs, uint64_t stride, uint64_t count, uint64_t readCount, uint64_t* toc, uint32_t* values)
{
    HIP_CHECK(hipSetDevice(device));
    if(count >= (1ULL << 32)) throw std::runtime_error("pair_table: too many pairs (the table's indices are 32-bit, as the reference's sort keys are).");
    if(stride < sizeof(shasta_oriented_read_pair) || stride % 4 != 0) throw std::runtime_error("pair_table: bad stride.");
    const uint64_t rows = 2 * readCount, n = 4 * count;
    if(count == 0) { std::fill(toc, toc + rows + 1, uint64_t(0)); return; }
    const int otherBits = bitsFor(std::max<uint64_t>(rows, 2));
    if(2 * otherBits > 64) throw std::runtime_error("pair_table: too many reads.");
    const ScopedStream scopedStream;                       // (destroyed on every path, a throwing HIP_CHECK included)
    hipStream_t stream = scopedStream;
    // Entries one radix sort takes (2^31: 32-bit positions, and 56 GB of keys, values and their doubles); beyond it the table
    // is built range by range.  SHASTA_MI355X_PAIR_TABLE_KEYS lowers it (tests: the ranges at a size a test can hold).
    const uint64_t keyLimit = [] {
        const char* e = std::getenv("SHASTA_MI355X_PAIR_TABLE_KEYS");
        return e ? std::min<uint64_t>(std::max<uint64_t>(std::strtoull(e, nullptr, 10), 4), 1ULL << 31) : (1ULL << 31);
    }();
    if(n > keyLimit) { pairTableInRanges(static_cast<const uint8_t*>(pairs), stride, count, rows, otherBits, keyLimit, toc, values, stream); return; }

    DeviceBuffer<uint8_t> devicePairs;
    DeviceBuffer<uint64_t> keysA, keysB, deviceToc;
    DeviceBuffer<uint32_t> valuesA, valuesB, bad;
    RadixSortWorkspace ws;
    devicePairs.reserve(count * stride, stream); keysA.reserve(n, stream); keysB.reserve(n, stream);
    valuesA.reserve(n, stream); valuesB.reserve(n, stream); deviceToc.reserve(rows + 1, stream); bad.reserve(1, stream);
    HIP_CHECK(hipMemcpyAsync(devicePairs.data(), pairs, count * stride, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemsetAsync(bad.data(), 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(pairTableKeysKernel, dim3(divUp(count, 256)), dim3(256), 0, stream,
        (const uint8_t*)devicePairs.data(), stride, count, rows, otherBits, keysA.data(), valuesA.data(), bad.data());
    HIP_CHECK(hipGetLastError());
    // Stable: equal (oriented read, partner) keys keep ascending pair indices, the order std::sort gives the reference's
    // (OrientedReadId, index) pairs.
    const bool inB = radixSort<uint64_t, uint32_t, true>(keysA.data(), keysB.data(), valuesA.data(), valuesB.data(), n, 2 * otherBits, ws, stream);
    const uint64_t* sortedKeys = inB ? keysB.data() : keysA.data();
    const uint32_t* sortedValues = inB ? valuesB.data() : valuesA.data();
    hipLaunchKernelGGL(rowStartsKernel, dim3(divUp(rows + 1, 256)), dim3(256), 0, stream, sortedKeys, n, rows, otherBits, deviceToc.data());
    HIP_CHECK(hipGetLastError());
    uint32_t hostBad = 0;
    HIP_CHECK(hipMemcpyAsync(&hostBad, bad.data(), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(toc, deviceToc.data(), (rows + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(values, sortedValues, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));

    if(hostBad) throw std::runtime_error("pair_table: a pair names a read beyond readCount.");
}

// Assembler::computeAlignmentTable (src/AssemblerAlign.cpp:509-571) as the last step of computeAlignments (:296): the table
// of the alignments the context's last borrowed aligner call stored.  Everything it needs is kept by the context from call
// to call (nothing is allocated from the second call on): the 12-byte heads of the rows {readId0, readId1, isSameStrand}
// go up from a page-locked array (a quarter of the bytes of the rows themselves), one stable sort of the 4 N keys on the
// context's stream, the table comes back into page-locked arrays that the caller reads in place.
namespace {
struct TableStore {
    PinnedBuffer heads, toc, values;
    DeviceBuffer<uint8_t> deviceHeads;
    DeviceBuffer<uint64_t> keysA, keysB, deviceToc;
    DeviceBuffer<uint32_t> valuesA, valuesB, bad;
    // The keys of the rows that the aligner's workers handed over while they were still on the device (alignmentTableKeysBegin /
    // alignmentTableKeysOfBatch): rows [0, keyedRows) when every batch of the call did, in any order, without a gap.
    std::atomic<uint64_t> keyedRows{0};
    uint64_t keyedCapacity = 0;
    int keyedOtherBits = 0;
};

// pairTableKeysKernel for rows firstRow ... of the table (the rows of one batch, still in the worker's device buffer).
__global__ void __launch_bounds__(256)
pairTableKeysOfRowsKernel(const shasta_alignment_data* __restrict__ rows, uint64_t count, uint64_t firstRow, uint64_t orientedReadCount, int otherBits,
    uint64_t* __restrict__ keys, uint32_t* __restrict__ values, uint32_t* __restrict__ bad)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if(i >= count) return;
    const shasta_oriented_read_pair pair = rows[i].pair;
    const uint64_t o0 = uint64_t(pair.readIds[0]) << 1, o1 = (uint64_t(pair.readIds[1]) << 1) | (pair.isSameStrand ? 0u : 1u);
    if(o0 >= orientedReadCount || o1 >= orientedReadCount) { atomicAdd(bad, 1u); }
    const uint64_t tableRows[4] = {o0, o1, o0 ^ 1u, o1 ^ 1u}, others[4] = {o1, o0, o1 ^ 1u, o0 ^ 1u};
#pragma unroll
    for(int k = 0; k < 4; k++) { keys[4 * (firstRow + i) + k] = (tableRows[k] << otherBits) | others[k]; values[4 * (firstRow + i) + k] = uint32_t(firstRow + i); }
}
}  // namespace

// A borrowed aligner call begins: room for the keys of up to maxRows alignments (nothing in flight on the context).
void alignmentTableKeysBegin(Context& ctx, uint64_t maxRows)
{
    if(!ctx.tableStore) ctx.tableStore = std::make_shared<TableStore>();
    TableStore& t = *static_cast<TableStore*>(ctx.tableStore.get());
    t.keyedRows.store(0); t.keyedCapacity = 0;
    if(maxRows == 0 || maxRows >= (1ULL << 29)) return;
    const uint64_t n = 4 * maxRows;
    t.keysA.reserve(n, ctx.stream); t.keysB.reserve(n, ctx.stream); t.valuesA.reserve(n, ctx.stream); t.valuesB.reserve(n, ctx.stream); t.bad.reserve(1, ctx.stream);
    HIP_CHECK(hipMemsetAsync(t.bad.data(), 0, sizeof(uint32_t), ctx.stream));
    HIP_CHECK(hipStreamSynchronize(ctx.stream));
    t.keyedCapacity = maxRows;
    t.keyedOtherBits = bitsFor(std::max<uint64_t>(2 * ctx.readCount, 2));
}

// A batch whose place in the call's results is known hands its rows over from the worker's device buffer, on the worker's stream:
// the table's keys of those rows are written where they belong, and the rows need not come back up for the table.
void alignmentTableKeysOfBatch(Context& ctx, const shasta_alignment_data* deviceRows, uint64_t count, uint64_t firstRow, hipStream_t stream)
{
    TableStore* t = static_cast<TableStore*>(ctx.tableStore.get());
    if(!t || count == 0 || firstRow + count > t->keyedCapacity) return;
    hipLaunchKernelGGL(pairTableKeysOfRowsKernel, dim3(divUp(count, 256)), dim3(256), 0, stream,
        deviceRows, count, firstRow, 2 * ctx.readCount, t->keyedOtherBits, t->keysA.data(), t->valuesA.data(), t->bad.data());
    HIP_CHECK(hipGetLastError());
    t->keyedRows.fetch_add(count);
}

void alignmentTableOfLastCall(Context& ctx, const uint64_t** tocOut, const uint32_t** valuesOut, uint64_t* valueCount)
{
    HIP_CHECK(hipSetDevice(ctx.device));
    uint64_t count = 0;
    const shasta_alignment_data* rows = borrowedAlignmentRows(ctx, &count);
    if(count >= (1ULL << 29)) throw std::runtime_error("alignment_table: 2^29 alignments or more (shasta_mi355x_pair_table builds such a table range by range).");
    if(!ctx.tableStore) ctx.tableStore = std::make_shared<TableStore>();
    TableStore& t = *static_cast<TableStore*>(ctx.tableStore.get());
    hipStream_t stream = ctx.stream;
    const uint64_t tableRows = 2 * ctx.readCount, n = 4 * count;
    const int otherBits = bitsFor(std::max<uint64_t>(tableRows, 2));
    uint64_t* toc = static_cast<uint64_t*>(t.toc.reserve((tableRows + 1) * sizeof(uint64_t)));
    uint32_t* values = static_cast<uint32_t*>(t.values.reserve(std::max<uint64_t>(1, n) * sizeof(uint32_t)));
    *tocOut = toc; *valuesOut = values; *valueCount = n;
    if(count == 0) { std::fill(toc, toc + tableRows + 1, uint64_t(0)); return; }
    constexpr uint64_t stride = sizeof(shasta_oriented_read_pair);
    // Every batch of the call handed its rows over on the device: the keys are there already.
    const bool keyed = t.keyedCapacity >= count && t.keyedRows.load() == count && t.keyedOtherBits == otherBits;
    t.keyedRows.store(~0ULL);           // (the sort below consumes the keys: a second table of the same call takes the rows from the host)
    static const bool debug = std::getenv("SHASTA_MI355X_DEBUG") != nullptr;
    if(debug) std::fprintf(stderr, "alignment table: %llu alignments, keys %s\n", (unsigned long long)count, keyed ? "left on the device by the batches" : "from the rows on the host");
    uint8_t* heads = keyed ? nullptr : static_cast<uint8_t*>(t.heads.reserve(count * stride));
    if(!keyed) {   // (the rows are 64 bytes apart in ordinary memory: a few host threads, each a contiguous share)
        const uint64_t threads = std::min<uint64_t>(4, std::max<uint64_t>(1, count >> 16));
        auto share = [&](uint64_t k) {
            for(uint64_t i = count * k / threads; i < count * (k + 1) / threads; i++) std::memcpy(heads + i * stride, &rows[i].pair, stride);
        };
        std::vector<std::thread> others;
        for(uint64_t k = 1; k < threads; k++) others.emplace_back(share, k);
        share(0);
        for(std::thread& o : others) o.join();
    }
    if(!keyed) {
        t.deviceHeads.reserve(count * stride, stream); t.keysA.reserve(n, stream); t.keysB.reserve(n, stream);
        t.valuesA.reserve(n, stream); t.valuesB.reserve(n, stream); t.bad.reserve(1, stream);
    }
    t.deviceToc.reserve(tableRows + 1, stream);
    const KernelTimers::Span span = ctx.timers.begin("alignment table (keys, sort, row starts)", stream);
    if(!keyed) {
        HIP_CHECK(hipMemcpyAsync(t.deviceHeads.data(), heads, count * stride, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemsetAsync(t.bad.data(), 0, sizeof(uint32_t), stream));
        hipLaunchKernelGGL(pairTableKeysKernel, dim3(divUp(count, 256)), dim3(256), 0, stream,
            (const uint8_t*)t.deviceHeads.data(), stride, count, tableRows, otherBits, t.keysA.data(), t.valuesA.data(), t.bad.data());
        HIP_CHECK(hipGetLastError());
    }
    const bool inB = radixSort<uint64_t, uint32_t, true>(t.keysA.data(), t.keysB.data(), t.valuesA.data(), t.valuesB.data(), n, 2 * otherBits, ctx.sortWs, stream);
    hipLaunchKernelGGL(rowStartsKernel, dim3(divUp(tableRows + 1, 256)), dim3(256), 0, stream,
        (const uint64_t*)(inB ? t.keysB.data() : t.keysA.data()), n, tableRows, otherBits, t.deviceToc.data());
    HIP_CHECK(hipGetLastError());
    // Booked: the heads read, the keys and values written, and one read + write of both per sorting pass.
    (void)ctx.timers.end(span, count * stride + 12 * n * (1 + 2 * uint64_t((2 * otherBits + 7) / 8)), count);
    uint32_t hostBad = 0;
    HIP_CHECK(hipMemcpyAsync(&hostBad, t.bad.data(), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(toc, t.deviceToc.data(), (tableRows + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(values, inB ? t.valuesB.data() : t.valuesA.data(), n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if(hostBad) throw std::runtime_error("alignment_table: an alignment names a read beyond readCount.");
}

// keep: uint8[count], 1 where the alignment stays in the read graph.
void readGraphKeep(int device, const shasta_alignment_data* alignmentData, uint64_t count, uint64_t readCount, uint32_t maxAlignmentCount, uint8_t* keep)
{
    HIP_CHECK(hipSetDevice(device));
    if(count >= (1ULL << 31)) throw std::runtime_error("read_graph_keep: too many alignments.");
    if(count == 0) return;
    const uint64_t n = 2 * count;
    const ScopedStream scopedStream;                       // (destroyed on every path, a throwing HIP_CHECK included)
    hipStream_t stream = scopedStream;

    DeviceBuffer<shasta_alignment_data> rows;
    DeviceBuffer<uint64_t> qualityA, qualityB;
    DeviceBuffer<uint32_t> reads, ids, entryA, entryB, readA, readB, bad;
    DeviceBuffer<uint8_t> deviceKeep;
    RadixSortWorkspace ws;
    rows.reserve(count, stream); qualityA.reserve(n, stream); qualityB.reserve(n, stream); reads.reserve(n, stream); ids.reserve(n, stream);
    entryA.reserve(n, stream); entryB.reserve(n, stream); readA.reserve(n, stream); readB.reserve(n, stream); bad.reserve(1, stream);
    deviceKeep.reserve(count, stream);
    HIP_CHECK(hipMemcpyAsync(rows.data(), alignmentData, count * sizeof(shasta_alignment_data), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemsetAsync(bad.data(), 0, sizeof(uint32_t), stream));
    HIP_CHECK(hipMemsetAsync(deviceKeep.data(), 0, count, stream));
    hipLaunchKernelGGL(readGraphKeysKernel, dim3(divUp(count, 256)), dim3(256), 0, stream,
        (const shasta_alignment_data*)rows.data(), count, readCount, qualityA.data(), reads.data(), ids.data(), bad.data());
    hipLaunchKernelGGL(fillIotaKernel, dim3(divUp(n, 256)), dim3(256), 0, stream, entryA.data(), n);
    HIP_CHECK(hipGetLastError());
    // By quality (best first), then -- stably -- by read.
    const bool firstInB = radixSort<uint64_t, uint32_t, true>(qualityA.data(), qualityB.data(), entryA.data(), entryB.data(), n, 64, ws, stream);
    uint32_t* byQuality = firstInB ? entryB.data() : entryA.data();
    uint32_t* spare = firstInB ? entryA.data() : entryB.data();
    hipLaunchKernelGGL(gatherReadsKernel, dim3(divUp(n, 256)), dim3(256), 0, stream, (const uint32_t*)byQuality, (const uint32_t*)reads.data(), n, readA.data());
    HIP_CHECK(hipGetLastError());
    const bool secondInB = radixSort<uint32_t, uint32_t, true>(readA.data(), readB.data(), byQuality, spare, n, bitsFor(std::max<uint64_t>(readCount, 2)), ws, stream);
    const uint32_t* sortedReads = secondInB ? readB.data() : readA.data();
    const uint32_t* sortedEntries = secondInB ? spare : byQuality;
    hipLaunchKernelGGL(readGraphKeepKernel, dim3(divUp(n, 256)), dim3(256), 0, stream, sortedReads, sortedEntries, (const uint32_t*)ids.data(), n, maxAlignmentCount, deviceKeep.data());
    HIP_CHECK(hipGetLastError());
    uint32_t hostBad = 0;
    HIP_CHECK(hipMemcpyAsync(&hostBad, bad.data(), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(keep, deviceKeep.data(), count, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));

    if(hostBad) throw std::runtime_error("read_graph_keep: an alignment names a read beyond readCount.");
}

}  // namespace shasta_mi355x
