// C ABI of libshasta_mi355x.so (include/shasta_mi355x.h).  Exceptions never cross
// the boundary: every entry point returns non-zero and records the message.
#include "context.hpp"

#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

#include <atomic>
#include <algorithm>
#include <mutex>
using namespace shasta_mi355x;

struct shasta_mi355x_ctx { Context impl; explicit shasta_mi355x_ctx(int d) : impl(d) {} };
struct shasta_mi355x_group { Group impl; shasta_mi355x_group(int n, const int* devices) : impl(n, devices) {} };

static thread_local std::string lastError;

namespace shasta_mi355x {
__global__ void __launch_bounds__(256) scrambleKernel(uint32_t* __restrict__ words, uint64_t count, uint64_t seed)
{
    for(uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += uint64_t(gridDim.x) * blockDim.x) {
        uint64_t x = (i + 1) * 0x9e3779b97f4a7c15ULL ^ seed;
        x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 32;
        words[i] = uint32_t(x);
    }
}
void scrambleDeviceMemory(void* p, size_t bytes, hipStream_t stream)
{
    static std::atomic<uint64_t> counter(0);
    const uint64_t count = bytes / 4;
    if(count == 0) return;
    const uint64_t seed = (counter.fetch_add(1) + 1) * 0xd6e8feb86659fd93ULL;
    hipLaunchKernelGGL(scrambleKernel, dim3(unsigned(std::min<uint64_t>((count + 255) / 256, 4096))), dim3(256), 0, stream, static_cast<uint32_t*>(p), count, seed);
    HIP_CHECK(hipGetLastError());
}
}  // namespace shasta_mi355x

// An aligner call keeps six workers' streams and their side streams busy; the HIP runtime deals a process's streams to
// GPU_MAX_HW_QUEUES hardware queues -- four unless the environment says otherwise -- and streams that share a queue run one after
// the other (round 3: 185 ms per aligner call with four queues against 158 with eight when other libraries had created streams
// first).  The library does NOT touch the environment (round 4's constructor called setenv at load: a write to the host process's
// environment that races with getenv in a threaded host such as Shasta, and does nothing when the runtime is already up): the
// caller sets GPU_MAX_HW_QUEUES=8 before its first HIP call (shasta_amd/lib.py and bench.py do), and a context
// created while the variable is unset says so once on stderr.
static void noteHardwareQueuesOnce()
{
    static std::once_flag once;
    std::call_once(once, [] {
        if(!std::getenv("GPU_MAX_HW_QUEUES") && !std::getenv("SHASTA_MI355X_QUIET"))
            std::fprintf(stderr, "shasta_mi355x: GPU_MAX_HW_QUEUES is not set; set it to 8 before the first HIP call of the process "
                "(the aligner's streams share four hardware queues otherwise: about 15 %% slower; INTEGRATION.md)\n");
    });
}

#define API_BEGIN try {
#define API_END(rc) } catch(const std::exception& e) { lastError = e.what(); return rc; } \
                      catch(...) { lastError = "unknown error"; return rc; }

extern "C" {

const char* shasta_mi355x_last_error(void) { return lastError.c_str(); }
const char* shasta_mi355x_version(void) { return "shasta_mi355x 0.1 (gfx950)"; }

// What in the process's environment costs the library speed (the stderr note of noteHardwareQueuesOnce, as a value an integrator can test):
// bit 0: GPU_MAX_HW_QUEUES is not set (the aligner's streams share four hardware queues: about 15 % slower).
int shasta_mi355x_environment_warnings(void) { return std::getenv("GPU_MAX_HW_QUEUES") ? 0 : 1; }

int shasta_mi355x_device_count(void)
{
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) return 0;
    int usable = 0;
    for(int i = 0; i < n; i++) {
        hipDeviceProp_t prop;
        if(hipGetDeviceProperties(&prop, i) == hipSuccess && std::string(prop.gcnArchName).rfind("gfx950", 0) == 0) ++usable;
    }
    return usable;
}

shasta_mi355x_ctx* shasta_mi355x_create(int device)
{
    API_BEGIN
    noteHardwareQueuesOnce();
    return new shasta_mi355x_ctx(device);
    API_END(nullptr)
}

void shasta_mi355x_destroy(shasta_mi355x_ctx* c) { delete c; }

int shasta_mi355x_set_markers(shasta_mi355x_ctx* c, uint64_t readCount,
    const uint64_t* markersToc, const void* markersData, const uint8_t* readFlags)
{
    API_BEGIN
    if(!c || !markersToc || (!markersData && markersToc[2 * readCount])) throw std::runtime_error("set_markers: null argument");
    c->impl.setMarkers(readCount, markersToc, markersData, nullptr, readFlags);
    return 0;
    API_END(1)
}

int shasta_mi355x_set_kmer_ids(shasta_mi355x_ctx* c, uint64_t readCount,
    const uint64_t* markersToc, const uint32_t* kmerIds, const uint8_t* readFlags)
{
    API_BEGIN
    if(!c || !markersToc || (!kmerIds && markersToc[2 * readCount])) throw std::runtime_error("set_kmer_ids: null argument");
    c->impl.setMarkers(readCount, markersToc, nullptr, kmerIds, readFlags);
    return 0;
    API_END(1)
}

int shasta_mi355x_set_kmer_ids_device(shasta_mi355x_ctx* c, uint64_t readCount,
    const uint64_t* markersToc, const void* kmerIdsDevice, const uint8_t* readFlags)
{
    API_BEGIN
    if(!c || !markersToc || (!kmerIdsDevice && markersToc[2 * readCount])) throw std::runtime_error("set_kmer_ids_device: null argument");
    c->impl.setMarkers(readCount, markersToc, nullptr, static_cast<const uint32_t*>(kmerIdsDevice), readFlags, true);
    return 0;
    API_END(1)
}

int shasta_mi355x_memcpy(shasta_mi355x_ctx* c, void* dst, const void* src, uint64_t bytes, int kind)
{
    API_BEGIN
    if(!c) throw std::runtime_error("memcpy: null context");
    HIP_CHECK(hipSetDevice(c->impl.device));
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
    if(bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, k, c->impl.stream));
    HIP_CHECK(hipStreamSynchronize(c->impl.stream));
    return 0;
    API_END(1)
}

int shasta_mi355x_lh_begin(shasta_mi355x_ctx* c, const shasta_lowhash0_params* params, int rank, int worldSize,
    const uint64_t* readBoundaries, uint32_t* log2BucketCount)
{
    API_BEGIN
    if(!c || !params || !readBoundaries) throw std::runtime_error("lh_begin: null argument");
    lowhash0Begin(c->impl, *params, rank, worldSize, readBoundaries, log2BucketCount);
    return 0;
    API_END(1)
}

int shasta_mi355x_lh_one_pass_fits(shasta_mi355x_ctx* c, int* fits)
{
    API_BEGIN
    if(!c || !fits) throw std::runtime_error("lh_one_pass_fits: null argument");
    *fits = lowhash0OnePassFits(c->impl) ? 1 : 0;
    return 0;
    API_END(1)
}

int shasta_mi355x_lh_hash(shasta_mi355x_ctx* c, uint64_t iteration, uint64_t* sendOffsets, const void** keysDevice, const void** valsDevice)
{
    API_BEGIN
    if(!c || !sendOffsets || !keysDevice || !valsDevice) throw std::runtime_error("lh_hash: null argument");
    const uint32_t* k = nullptr; const uint64_t* v = nullptr;
    lowhash0Hash(c->impl, iteration, sendOffsets, &k, &v);
    *keysDevice = k; *valsDevice = v;
    return 0;
    API_END(1)
}

int shasta_mi355x_lh_buckets(shasta_mi355x_ctx* c, const void* keysDevice, const void* valsDevice, uint64_t n,
    uint64_t* sendOffsets, const void** pairKeysDevice, uint64_t* bucketsUsed,
    uint64_t* sizeHistogram, uint32_t* overflowSizes, uint64_t overflowCapacity, uint64_t* overflowCount)
{
    API_BEGIN
    if(!c || !sendOffsets || !pairKeysDevice || !bucketsUsed || !sizeHistogram || !overflowCount) throw std::runtime_error("lh_buckets: null argument");
    const uint64_t* pk = nullptr;
    std::vector<uint32_t> overflow;
    lowhash0Buckets(c->impl, static_cast<const uint32_t*>(keysDevice), static_cast<const uint64_t*>(valsDevice), n,
        sendOffsets, &pk, bucketsUsed, sizeHistogram, overflow);
    if(overflow.size() > overflowCapacity) throw std::runtime_error("lh_buckets: overflow list capacity too small");
    if(!overflow.empty()) std::memcpy(overflowSizes, overflow.data(), overflow.size() * 4);
    *overflowCount = overflow.size();
    *pairKeysDevice = pk;
    return 0;
    API_END(1)
}

int shasta_mi355x_lh_merge(shasta_mi355x_ctx* c, const void* pairKeysDevice, uint64_t n, int evaluateNow,
    uint64_t* highFrequency, uint64_t* total)
{
    API_BEGIN
    if(!c || !highFrequency || !total) throw std::runtime_error("lh_merge: null argument");
    lowhash0Merge(c->impl, static_cast<const uint64_t*>(pairKeysDevice), n, evaluateNow != 0, highFrequency, total);
    return 0;
    API_END(1)
}

int shasta_mi355x_lh_hash_all(shasta_mi355x_ctx* c, uint64_t* sendOffsets, const void** keysDevice, const void** valsDevice)
{
    API_BEGIN
    if(!c || !sendOffsets || !keysDevice || !valsDevice) throw std::runtime_error("lh_hash_all: null argument");
    const uint64_t* k = nullptr; const uint64_t* v = nullptr;
    lowhash0HashAll(c->impl, sendOffsets, &k, &v);
    *keysDevice = k; *valsDevice = v;
    return 0;
    API_END(1)
}

int shasta_mi355x_lh_buckets_all(shasta_mi355x_ctx* c, const void* keysDevice, const void* valsDevice, uint64_t n,
    uint64_t* sendOffsets, const void** pairKeysDevice, const void** pairTagsDevice, uint64_t iterationCapacity, uint64_t* bucketsUsed,
    uint64_t* sizeHistogram, uint64_t* overflowSizes, uint64_t overflowCapacity, uint64_t* overflowCount)
{
    API_BEGIN
    if(!c || !sendOffsets || !pairKeysDevice || !pairTagsDevice || !bucketsUsed || !sizeHistogram || !overflowCount) throw std::runtime_error("lh_buckets_all: null argument");
    if(iterationCapacity < lowhash0JobPlannedIterations(c->impl)) throw std::runtime_error("lh_buckets_all: iteration capacity too small");
    const uint64_t* pk = nullptr; const uint32_t* tags = nullptr;
    std::vector<uint64_t> overflow;
    lowhash0BucketsAll(c->impl, static_cast<const uint64_t*>(keysDevice), static_cast<const uint64_t*>(valsDevice), n,
        sendOffsets, &pk, &tags, bucketsUsed, sizeHistogram, overflow);
    if(overflow.size() > overflowCapacity) throw std::runtime_error("lh_buckets_all: overflow list capacity too small");
    if(!overflow.empty()) std::memcpy(overflowSizes, overflow.data(), overflow.size() * 8);
    *overflowCount = overflow.size();
    *pairKeysDevice = pk; *pairTagsDevice = tags;
    return 0;
    API_END(1)
}

int shasta_mi355x_lh_merge_all(shasta_mi355x_ctx* c, const void* pairKeysDevice, const void* pairTagsDevice, uint64_t n)
{
    API_BEGIN
    if(!c) throw std::runtime_error("lh_merge_all: null argument");
    lowhash0MergeAll(c->impl, static_cast<const uint64_t*>(pairKeysDevice), static_cast<const uint32_t*>(pairTagsDevice), n);
    return 0;
    API_END(1)
}

int shasta_mi355x_lh_finish(shasta_mi355x_ctx* c, uint64_t* readLowHashStatistics,
    shasta_oriented_read_pair** candidates, uint64_t* candidateCount,
    uint64_t* highFrequencyPerIteration, uint64_t* totalPerIteration, uint64_t iterationCapacity, uint64_t* iterationCount)
{
    API_BEGIN
    if(!c || !readLowHashStatistics || !candidates || !candidateCount || !iterationCount) throw std::runtime_error("lh_finish: null argument");
    // Checked BEFORE the job is retired (lowhash0Finish discards it): a capacity that is too small, or arrays that are missing,
    // must not cost the caller the job's result.
    const uint64_t iterationsOfJob = lowhash0JobIterations(c->impl);
    if(iterationsOfJob > iterationCapacity) throw std::runtime_error("lh_finish: iteration capacity too small (the job has " + std::to_string(iterationsOfJob) + " iterations; nothing was discarded, call again with room)");
    if(iterationsOfJob && (!highFrequencyPerIteration || !totalPerIteration)) throw std::runtime_error("lh_finish: null per-iteration array");
    std::vector<shasta_oriented_read_pair> v;
    std::vector<uint64_t> high, total;
    lowhash0Finish(c->impl, readLowHashStatistics, v, high, total);
    if(high.size() > iterationCapacity) throw std::runtime_error("lh_finish: iteration capacity too small");
    for(size_t k = 0; k < high.size(); k++) { highFrequencyPerIteration[k] = high[k]; totalPerIteration[k] = total[k]; }
    *iterationCount = high.size();
    shasta_oriented_read_pair* p = static_cast<shasta_oriented_read_pair*>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(shasta_oriented_read_pair)));
    if(!p) throw std::bad_alloc();
    if(!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(shasta_oriented_read_pair));
    *candidates = p; *candidateCount = v.size();
    return 0;
    API_END(1)
}

int shasta_mi355x_lh_finish_on_device(shasta_mi355x_ctx* c, uint64_t* readLowHashStatistics,
    const shasta_oriented_read_pair** candidatesDevice, uint64_t* candidateCount,
    uint64_t* highFrequencyPerIteration, uint64_t* totalPerIteration, uint64_t iterationCapacity, uint64_t* iterationCount)
{
    API_BEGIN
    if(!c || !readLowHashStatistics || !candidatesDevice || !candidateCount || !iterationCount) throw std::runtime_error("lh_finish_on_device: null argument");
    const uint64_t iterationsOfJob = lowhash0JobIterations(c->impl);            // (checked before the job is retired, as in lh_finish)
    if(iterationsOfJob > iterationCapacity) throw std::runtime_error("lh_finish_on_device: iteration capacity too small (the job has " + std::to_string(iterationsOfJob) + " iterations; nothing was discarded, call again with room)");
    if(iterationsOfJob && (!highFrequencyPerIteration || !totalPerIteration)) throw std::runtime_error("lh_finish_on_device: null per-iteration array");
    std::vector<uint64_t> high, total;
    lowhash0Finish(c->impl, readLowHashStatistics, nullptr, high, total, candidatesDevice, candidateCount);
    for(size_t k = 0; k < high.size(); k++) { highFrequencyPerIteration[k] = high[k]; totalPerIteration[k] = total[k]; }
    *iterationCount = high.size();
    return 0;
    API_END(1)
}

void shasta_mi355x_free(void* p) { std::free(p); }

int shasta_mi355x_lowhash0_run(shasta_mi355x_ctx* c, const shasta_lowhash0_params* params,
    uint64_t* readLowHashStatistics, shasta_lowhash0_result* result)
{
    API_BEGIN
    if(!c || !params || !readLowHashStatistics || !result) throw std::runtime_error("lowhash0_run: null argument");
    lowhash0Run(c->impl, *params, readLowHashStatistics, *result);
    return 0;
    API_END(1)
}

int shasta_mi355x_lowhash0(uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    const uint8_t* readFlags, const shasta_lowhash0_params* params,
    uint64_t* readLowHashStatistics, shasta_lowhash0_result* result)
{
    API_BEGIN
    if(!markersToc || !params || !readLowHashStatistics || !result) throw std::runtime_error("lowhash0: null argument");
    int device = 0;
    (void)hipGetDevice(&device);
    shasta_mi355x_ctx c(device);
    c.impl.setMarkers(readCount, markersToc, markersData, nullptr, readFlags);
    lowhash0Run(c.impl, *params, readLowHashStatistics, *result);
    return 0;
    API_END(1)
}

void shasta_mi355x_lowhash0_free(shasta_lowhash0_result* r) { if(r) lowhash0Free(*r); }

int shasta_mi355x_align4_run(shasta_mi355x_ctx* c, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates, const shasta_align4_options* options,
    int wantOrdinals, shasta_align4_result* result)
{
    API_BEGIN
    if(!c || (!candidates && candidateCount) || !options || !result) throw std::runtime_error("align4_run: null argument");
    align4Run(c->impl, candidateCount, candidates, *options, wantOrdinals != 0, *result);
    return 0;
    API_END(1)
}

int shasta_mi355x_align4_run_borrowed(shasta_mi355x_ctx* c, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates, const shasta_align4_options* options,
    int wantOrdinals, shasta_align4_result* result)
{
    API_BEGIN
    if(!c || (!candidates && candidateCount) || !options || !result) throw std::runtime_error("align4_run_borrowed: null argument");
    align4Run(c->impl, candidateCount, candidates, *options, wantOrdinals != 0, *result, true);
    return 0;
    API_END(1)
}

int shasta_mi355x_align4_batch(uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align4_options* options, int wantOrdinals, shasta_align4_result* result)
{
    API_BEGIN
    if(!markersToc || (!candidates && candidateCount) || !options || !result) throw std::runtime_error("align4_batch: null argument");
    int device = 0;
    (void)hipGetDevice(&device);
    shasta_mi355x_ctx c(device);
    c.impl.setMarkers(readCount, markersToc, markersData, nullptr, nullptr);
    align4Run(c.impl, candidateCount, candidates, *options, wantOrdinals != 0, *result);
    return 0;
    API_END(1)
}

void shasta_mi355x_align4_free(shasta_align4_result* r) { if(r) align4Free(*r); }

int shasta_mi355x_align3_run(shasta_mi355x_ctx* c, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates, const shasta_align3_options* options,
    int wantOrdinals, int borrowed, shasta_align4_result* result)
{
    API_BEGIN
    if(!c || (!candidates && candidateCount) || !options || !result) throw std::runtime_error("align3_run: null argument");
    align3Run(c->impl, candidateCount, candidates, *options, wantOrdinals != 0, *result, borrowed != 0);
    return 0;
    API_END(1)
}

int shasta_mi355x_align3_batch(uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align3_options* options, int wantOrdinals, shasta_align4_result* result)
{
    API_BEGIN
    if(!markersToc || (!candidates && candidateCount) || !options || !result) throw std::runtime_error("align3_batch: null argument");
    int device = 0;
    (void)hipGetDevice(&device);
    shasta_mi355x_ctx c(device);
    c.impl.setMarkers(readCount, markersToc, markersData, nullptr, nullptr);
    align3Run(c.impl, candidateCount, candidates, *options, wantOrdinals != 0, *result);
    return 0;
    API_END(1)
}

int shasta_mi355x_find_markers(shasta_mi355x_ctx* c, uint64_t readCount,
    const uint64_t* readsToc, const uint64_t* readsData, const uint64_t* baseCounts,
    uint64_t k, const void* kmerTable, uint64_t kmerInfoStride, uint64_t isMarkerOffset,
    const uint8_t* readFlags, int wantPacked, shasta_markers_result* result)
{
    API_BEGIN
    if(!readsToc || (!readsData && readsToc[readCount]) || (!baseCounts && readCount) || !kmerTable || !result) {
        throw std::runtime_error("find_markers: null argument");
    }
    if(c) {
        findMarkers(c->impl, readCount, readsToc, readsData, baseCounts, k, kmerTable, kmerInfoStride, isMarkerOffset, readFlags, wantPacked != 0, *result);
    } else {
        int device = 0;
        (void)hipGetDevice(&device);
        shasta_mi355x_ctx temporary(device);
        findMarkers(temporary.impl, readCount, readsToc, readsData, baseCounts, k, kmerTable, kmerInfoStride, isMarkerOffset, readFlags, wantPacked != 0, *result);
    }
    return 0;
    API_END(1)
}

void shasta_mi355x_find_markers_free(shasta_markers_result* r) { if(r) findMarkersFree(*r); }

int shasta_mi355x_kernel_table(shasta_mi355x_ctx* c, shasta_mi355x_kernel_stat* rows, uint64_t capacity, uint64_t* count)
{
    API_BEGIN
    if(!c || !count || (capacity && !rows)) throw std::runtime_error("kernel_table: null argument");
    const std::vector<KernelTimers::Entry> table = c->impl.timers.table();
    uint64_t n = 0;
    for(const KernelTimers::Entry& e : table) {
        if(e.launches == 0) continue;
        if(n < capacity) {
            shasta_mi355x_kernel_stat& r = rows[n];
            std::memset(&r, 0, sizeof(r));
            std::strncpy(r.name, e.name.c_str(), sizeof(r.name) - 1);
            r.seconds = e.seconds; r.launches = e.launches; r.algorithmicBytes = e.bytes; r.work = e.work;
        }
        ++n;
    }
    *count = n;
    return 0;
    API_END(1)
}

int shasta_mi355x_kernel_table_reset(shasta_mi355x_ctx* c)
{
    API_BEGIN
    if(!c) throw std::runtime_error("kernel_table_reset: null argument");
    c->impl.timers.reset();
    return 0;
    API_END(1)
}

int shasta_mi355x_calibrate(uint64_t bytes, int mode)
{
    API_BEGIN
    calibrateUnit(bytes, mode);
    return 0;
    API_END(1)
}

int shasta_mi355x_hash_windows(const uint32_t* kmerIds, uint64_t n, uint64_t m, uint64_t iteration, uint64_t* out)
{
    API_BEGIN
    hashWindowsUnit(kmerIds, n, m, iteration, out);
    return 0;
    API_END(1)
}

int shasta_mi355x_banded_dp(const uint32_t* k0, uint32_t nx, const uint32_t* k1, uint32_t ny,
    int32_t bandMin, int32_t bandMax, uint32_t* ordinals, uint64_t capacity, uint64_t* count, int32_t* score)
{
    API_BEGIN
    bandedDpUnit(k0, nx, k1, ny, bandMin, bandMax, ordinals, capacity, count, score);
    return 0;
    API_END(1)
}

int shasta_mi355x_banded_dp_many(const uint32_t* kmerIds, uint64_t kmerCount, uint64_t taskCount,
    const uint64_t* begin0, const uint32_t* nx, const uint64_t* begin1, const uint32_t* ny, const int32_t* bandMin, const int32_t* bandMax,
    uint64_t* counts, int32_t* scores, uint32_t* ordinals, uint64_t capacity, double* seconds, uint64_t* cells)
{
    API_BEGIN
    bandedDpManyUnit(kmerIds, kmerCount, taskCount, begin0, nx, begin1, ny, bandMin, bandMax, counts, scores, ordinals, capacity, seconds, cells);
    return 0;
    API_END(1)
}

int shasta_mi355x_palindromic_screen(shasta_mi355x_ctx* c, uint64_t deltaThreshold, uint32_t* bound)
{
    API_BEGIN
    if(!c || !bound) throw std::runtime_error("palindromic_screen: null argument");
    palindromicScreen(c->impl, deltaThreshold, bound);
    return 0;
    API_END(1)
}

int shasta_mi355x_pair_table(int device, const void* pairs, uint64_t strideBytes, uint64_t pairCount, uint64_t readCount, uint64_t* toc, uint32_t* values)
{
    API_BEGIN
    if((pairCount && !pairs) || !toc || (pairCount && !values)) throw std::runtime_error("pair_table: null argument");
    pairTable(device, pairs, strideBytes, pairCount, readCount, toc, values);
    return 0;
    API_END(1)
}

int shasta_mi355x_alignment_table(shasta_mi355x_ctx* c, const uint64_t** toc, const uint32_t** values, uint64_t* valueCount)
{
    API_BEGIN
    if(!c || !toc || !values || !valueCount) throw std::runtime_error("alignment_table: null argument");
    alignmentTableOfLastCall(c->impl, toc, values, valueCount);
    return 0;
    API_END(1)
}

int shasta_mi355x_read_graph_keep(int device, const shasta_alignment_data* alignmentData, uint64_t alignmentCount, uint64_t readCount,
    uint32_t maxAlignmentCount, uint8_t* keep)
{
    API_BEGIN
    if(alignmentCount && (!alignmentData || !keep)) throw std::runtime_error("read_graph_keep: null argument");
    readGraphKeep(device, alignmentData, alignmentCount, readCount, maxAlignmentCount, keep);
    return 0;
    API_END(1)
}

shasta_mi355x_group* shasta_mi355x_group_create(int deviceCount, const int* devices)
{
    API_BEGIN
    noteHardwareQueuesOnce();
    return new shasta_mi355x_group(deviceCount, devices);
    API_END(nullptr)
}

void shasta_mi355x_group_destroy(shasta_mi355x_group* g) { delete g; }

int shasta_mi355x_group_set_markers(shasta_mi355x_group* g, uint64_t readCount,
    const uint64_t* markersToc, const void* markersData, const uint8_t* readFlags)
{
    API_BEGIN
    if(!g || !markersToc || (!markersData && markersToc[2 * readCount])) throw std::runtime_error("group_set_markers: null argument");
    g->impl.setMarkers(readCount, markersToc, markersData, nullptr, readFlags);
    return 0;
    API_END(1)
}

int shasta_mi355x_group_set_kmer_ids(shasta_mi355x_group* g, uint64_t readCount,
    const uint64_t* markersToc, const uint32_t* kmerIds, const uint8_t* readFlags)
{
    API_BEGIN
    if(!g || !markersToc || (!kmerIds && markersToc[2 * readCount])) throw std::runtime_error("group_set_kmer_ids: null argument");
    g->impl.setMarkers(readCount, markersToc, nullptr, kmerIds, readFlags);
    return 0;
    API_END(1)
}

int shasta_mi355x_group_lowhash0_run(shasta_mi355x_group* g, const shasta_lowhash0_params* params,
    uint64_t* readLowHashStatistics, shasta_lowhash0_result* result)
{
    API_BEGIN
    if(!g || !params || !readLowHashStatistics || !result) throw std::runtime_error("group_lowhash0_run: null argument");
    g->impl.lowhash0Run(*params, readLowHashStatistics, *result);
    return 0;
    API_END(1)
}

int shasta_mi355x_group_align4_run(shasta_mi355x_group* g, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates, const shasta_align4_options* options, int wantOrdinals, shasta_align4_result* result)
{
    API_BEGIN
    if(!g || (!candidates && candidateCount) || !options || !result) throw std::runtime_error("group_align4_run: null argument");
    g->impl.alignRun(candidateCount, candidates, options, nullptr, wantOrdinals != 0, *result);
    return 0;
    API_END(1)
}

int shasta_mi355x_group_align3_run(shasta_mi355x_group* g, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates, const shasta_align3_options* options, int wantOrdinals, shasta_align4_result* result)
{
    API_BEGIN
    if(!g || (!candidates && candidateCount) || !options || !result) throw std::runtime_error("group_align3_run: null argument");
    g->impl.alignRun(candidateCount, candidates, nullptr, options, wantOrdinals != 0, *result);
    return 0;
    API_END(1)
}

int shasta_mi355x_group_align4_run_borrowed(shasta_mi355x_group* g, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates, const shasta_align4_options* options, int wantOrdinals, shasta_align4_result* result)
{
    API_BEGIN
    if(!g || (!candidates && candidateCount) || !options || !result) throw std::runtime_error("group_align4_run_borrowed: null argument");
    g->impl.alignRun(candidateCount, candidates, options, nullptr, wantOrdinals != 0, *result, true);
    return 0;
    API_END(1)
}

int shasta_mi355x_group_align3_run_borrowed(shasta_mi355x_group* g, uint64_t candidateCount,
    const shasta_oriented_read_pair* candidates, const shasta_align3_options* options, int wantOrdinals, shasta_align4_result* result)
{
    API_BEGIN
    if(!g || (!candidates && candidateCount) || !options || !result) throw std::runtime_error("group_align3_run_borrowed: null argument");
    g->impl.alignRun(candidateCount, candidates, nullptr, options, wantOrdinals != 0, *result, true);
    return 0;
    API_END(1)
}

int shasta_mi355x_lowhash0_multi(uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    const uint8_t* readFlags, const shasta_lowhash0_params* params, int deviceCount, const int* devices,
    uint64_t* readLowHashStatistics, shasta_lowhash0_result* result)
{
    API_BEGIN
    if(!markersToc || !params || !readLowHashStatistics || !result) throw std::runtime_error("lowhash0_multi: null argument");
    shasta_mi355x_group g(deviceCount, devices);
    g.impl.setMarkers(readCount, markersToc, markersData, nullptr, readFlags);
    g.impl.lowhash0Run(*params, readLowHashStatistics, *result);
    return 0;
    API_END(1)
}

int shasta_mi355x_align4_batch_multi(uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align4_options* options, int wantOrdinals, int deviceCount, const int* devices, shasta_align4_result* result)
{
    API_BEGIN
    if(!markersToc || (!candidates && candidateCount) || !options || !result) throw std::runtime_error("align4_batch_multi: null argument");
    shasta_mi355x_group g(deviceCount, devices);
    g.impl.setMarkers(readCount, markersToc, markersData, nullptr, nullptr);
    g.impl.alignRun(candidateCount, candidates, options, nullptr, wantOrdinals != 0, *result);
    return 0;
    API_END(1)
}

int shasta_mi355x_align3_batch_multi(uint64_t readCount, const uint64_t* markersToc, const void* markersData,
    uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align3_options* options, int wantOrdinals, int deviceCount, const int* devices, shasta_align4_result* result)
{
    API_BEGIN
    if(!markersToc || (!candidates && candidateCount) || !options || !result) throw std::runtime_error("align3_batch_multi: null argument");
    shasta_mi355x_group g(deviceCount, devices);
    g.impl.setMarkers(readCount, markersToc, markersData, nullptr, nullptr);
    g.impl.alignRun(candidateCount, candidates, nullptr, options, wantOrdinals != 0, *result);
    return 0;
    API_END(1)
}

}  // extern "C"
