// Several GPUs of one node behind ONE blocking call, from one process: the shape of the reference's two seams
// (Assembler::findAlignmentCandidatesLowHash0, /root/reference/src/AssemblerLowHash.cpp:36-52;
// Assembler::computeAlignments, src/AssemblerAlign.cpp:208-304 -- single C++ calls that fan out over threads inside).
// SURVEY 8(b) proposed `int nGpus` on both entry points; this is it.
//
// A group holds one context per device and one host thread per device for the duration of a call.  LowHash0 runs
// as the sharded job of SURVEY 8(e): every device holds all markers, hashes the reads of its own range (contiguous
// read-id ranges balanced by marker count), owns a contiguous range of bucket ids and the pair keys whose readId0 lies
// in its read range.  The two exchanges of an iteration are device-to-device copies over xGMI: after a host barrier
// every device PULLS its segment of every other device's sorted records / pair keys with hipMemcpyPeerAsync on its own
// stream (an all-to-all in which each of the 7 links of a device carries one segment); no staging through the host,
// no second process, no collective library.  The aligner needs no exchange: candidates are cut into contiguous ranges
// balanced by the markers they touch (sum of nx + ny), one range per device, results concatenated in candidate order.
// Reductions (per-read statistics, per-iteration counters, bucket histograms) happen on the host after the last
// iteration -- they are a few megabytes.
//
// Second transport, SHASTA_MI355X_GROUP_TRANSPORT=rccl: the same two exchanges as ONE grouped ncclSend / ncclRecv all-to-all each
// (RCCL over xGMI: one communicator per device from ncclCommInitAll, every device's host thread posts its sends and receives on its
// own stream) -- what BASELINE.json's north_star names for the redistribution of the bucket hits.  librccl.so is opened at run time
// (dlopen), so the library links against nothing but the HIP runtime and a machine without RCCL keeps the peer copies.  RCCL refuses a
// device listed twice, so the one-device tests of the group (device 0 three times) stay on peer copies; with ONE device and
// SHASTA_MI355X_GROUP_STAGED=1 the staged job runs with a world of one and RCCL sends to itself -- all a one-GPU box can exercise.
#include "context.hpp"
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

namespace shasta_mi355x {

namespace {

// A barrier for the device threads of one call that an error can break: a thread that failed calls abort(), and every
// thread that waits (now or later) gets an exception instead of a deadlock.
class CallBarrier {
public:
    explicit CallBarrier(int n) : count(n) {}
    void wait()
    {
        std::unique_lock<std::mutex> lock(mutex);
        if(aborted) throw std::runtime_error("another device of the group failed");
        const uint64_t myGeneration = generation;
        if(++waiting == count) { waiting = 0; ++generation; condition.notify_all(); return; }
        condition.wait(lock, [&] { return generation != myGeneration || aborted; });
        if(aborted && generation == myGeneration) throw std::runtime_error("another device of the group failed");
    }
    void abort() { std::lock_guard<std::mutex> lock(mutex); aborted = true; condition.notify_all(); }
private:
    std::mutex mutex;
    std::condition_variable condition;
    int count, waiting = 0;
    uint64_t generation = 0;
    bool aborted = false;
};

template<class T> T* mallocCopy(const std::vector<T>& v)
{
    T* p = static_cast<T*>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
    if(!p) throw std::bad_alloc();
    if(!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

// Runs fn(rank) on one thread per device; the first error (by rank) is rethrown after all threads have ended.
template<class F> void onEveryDevice(int world, CallBarrier& barrier, F fn)
{
    std::vector<std::string> errors;
    errors.resize(size_t(world));
    std::vector<std::thread> threads;
    auto body = [&](int rank) {
        try { fn(rank); }
        catch(const std::exception& e) { errors[size_t(rank)] = e.what()[0] ? e.what() : "error"; barrier.abort(); }
        catch(...) { errors[size_t(rank)] = "unknown error"; barrier.abort(); }
    };
    for(int r = 1; r < world; r++) threads.emplace_back(body, r);
    body(0);
    for(std::thread& t : threads) t.join();
    // Report the error that started it, not the "another device failed" it caused elsewhere.
    std::string first;
    for(const std::string& e : errors) if(!e.empty() && e != "another device of the group failed") { first = e; break; }
    if(first.empty()) for(const std::string& e : errors) if(!e.empty()) { first = e; break; }
    if(!first.empty()) throw std::runtime_error(first);
}

// librccl.so's entry points, resolved at run time.
struct RcclApi {
    void* handle = nullptr;
    int (*commInitAll)(void**, int, const int*) = nullptr;
    int (*commDestroy)(void*) = nullptr;
    int (*commAbort)(void*) = nullptr;
    int (*groupStart)() = nullptr;
    int (*groupEnd)() = nullptr;
    int (*send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*errorString)(int) = nullptr;
    bool ok() const { return handle && commInitAll && commDestroy && groupStart && groupEnd && send && recv; }
};
const RcclApi& rcclApi()
{
    static const RcclApi api = [] {
        RcclApi a;
        for(const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) { a.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL); if(a.handle) break; }
        if(!a.handle) return a;
        a.commInitAll = reinterpret_cast<decltype(a.commInitAll)>(dlsym(a.handle, "ncclCommInitAll"));
        a.commDestroy = reinterpret_cast<decltype(a.commDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
        a.commAbort = reinterpret_cast<decltype(a.commAbort)>(dlsym(a.handle, "ncclCommAbort"));
        a.groupStart = reinterpret_cast<decltype(a.groupStart)>(dlsym(a.handle, "ncclGroupStart"));
        a.groupEnd = reinterpret_cast<decltype(a.groupEnd)>(dlsym(a.handle, "ncclGroupEnd"));
        a.send = reinterpret_cast<decltype(a.send)>(dlsym(a.handle, "ncclSend"));
        a.recv = reinterpret_cast<decltype(a.recv)>(dlsym(a.handle, "ncclRecv"));
        a.errorString = reinterpret_cast<decltype(a.errorString)>(dlsym(a.handle, "ncclGetErrorString"));
        return a;
    }();
    return api;
}
void rcclCheck(int result, const char* what)
{
    if(result == 0) return;
    const RcclApi& api = rcclApi();
    throw std::runtime_error(std::string("RCCL error in ") + what + ": " + (api.errorString ? api.errorString(result) : std::to_string(result).c_str()));
}
constexpr int RCCL_UINT8 = 1;                 // ncclUint8 (rccl.h)
bool rcclTransportWanted() { const char* e = std::getenv("SHASTA_MI355X_GROUP_TRANSPORT"); return e && std::string(e) == "rccl"; }
// (test switch: the staged job with its exchanges even for a group of one device)
bool stagedWithOneDevice() { const char* e = std::getenv("SHASTA_MI355X_GROUP_STAGED"); return e && std::atoi(e) != 0; }

// The group's communicators, made at the first call that wants them; false: stay on peer copies (said once on stderr why).
bool ensureRccl(Group& group)
{
    if(!group.rcclCommunicators.empty()) return true;
    if(group.rcclTried) {
        // (an earlier call could not make the communicators, or one of its exchanges failed and they were aborted: said at every call, not once)
        std::fprintf(stderr, "shasta_mi355x group: SHASTA_MI355X_GROUP_TRANSPORT=rccl, but RCCL failed in an earlier call of this group; the exchanges are device-to-device copies\n");
        return false;
    }
    group.rcclTried = true;
    const RcclApi& api = rcclApi();
    std::vector<int> devices;
    for(const auto& c : group.contexts) devices.push_back(c->device);
    std::vector<int> sorted = devices;
    std::sort(sorted.begin(), sorted.end());
    const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
    if(!api.ok() || !distinct) {
        std::fprintf(stderr, "shasta_mi355x group: SHASTA_MI355X_GROUP_TRANSPORT=rccl, but %s; the exchanges stay device-to-device copies\n",
            !api.ok() ? "librccl.so could not be opened" : "a device is listed more than once (RCCL wants one rank per device)");
        return false;
    }
    std::vector<void*> communicators(devices.size(), nullptr);
    rcclCheck(api.commInitAll(communicators.data(), int(devices.size()), devices.data()), "ncclCommInitAll");
    group.rcclCommunicators = communicators;
    return true;
}

}  // namespace

Group::~Group()
{
    if(!rcclCommunicators.empty()) { const RcclApi& api = rcclApi(); for(void* c : rcclCommunicators) if(c && api.commDestroy) (void)api.commDestroy(c); }
}

Group::Group(int deviceCount, const int* devices)
{
    if(deviceCount < 1 || deviceCount > 64) throw std::runtime_error("shasta_mi355x group: device count must be in [1, 64].");
    for(int k = 0; k < deviceCount; k++) contexts.emplace_back(new Context(devices ? devices[k] : k));
    // Peer access between every pair of distinct devices (the exchanges are direct device-to-device copies).
    for(int a = 0; a < deviceCount; a++) {
        for(int b = 0; b < deviceCount; b++) {
            const int da = contexts[size_t(a)]->device, db = contexts[size_t(b)]->device;
            if(da == db) continue;
            int can = 0;
            HIP_CHECK(hipDeviceCanAccessPeer(&can, da, db));
            if(!can) throw std::runtime_error("shasta_mi355x group: device " + std::to_string(da) + " cannot access device " + std::to_string(db) + " (no xGMI / PCIe peer path).");
            HIP_CHECK(hipSetDevice(da));
            const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
            if(e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e);
            (void)hipGetLastError();
        }
    }
}

void Group::setMarkers(uint64_t readCount, const uint64_t* toc, const void* data7, const uint32_t* denseKmerIds, const uint8_t* flags)
{
    // The first device takes the host data (7-byte records are stripped there); the others copy its dense kmer ids
    // device to device: 4 bytes per marker over xGMI instead of 7 over PCIe, once per device.
    contexts[0]->setMarkers(readCount, toc, data7, denseKmerIds, flags);
    for(size_t k = 1; k < contexts.size(); k++) {
        contexts[k]->setMarkers(readCount, toc, nullptr, contexts[0]->kmerIds.data(), flags, true);
    }
}

// LowHash0::LowHash0 over the group.
void Group::lowhash0Run(const shasta_lowhash0_params& p, uint64_t* readLowHashStatistics, shasta_lowhash0_result& result)
{
    std::memset(&result, 0, sizeof(result));
    const auto t0 = std::chrono::steady_clock::now();
    const int world = int(contexts.size());
    if(world == 1 && !stagedWithOneDevice()) { shasta_mi355x::lowhash0Run(*contexts[0], p, readLowHashStatistics, result); return; }
    const bool overRccl = rcclTransportWanted() && ensureRccl(*this);
    const uint64_t readCount = contexts[0]->readCount;
    const std::vector<uint64_t>& toc = contexts[0]->hostToc;
    if(readCount == 0) throw std::runtime_error("LowHash0: no reads.");

    // Contiguous read ranges of about equal marker count.
    std::vector<uint64_t> boundaries(size_t(world) + 1, 0);
    {
        const uint64_t total = toc[2 * readCount];
        uint64_t read = 0;
        for(int r = 1; r < world; r++) {
            const uint64_t target = uint64_t(double(total) * double(r) / double(world));
            while(read < readCount && toc[2 * (read + 1)] <= target) ++read;
            boundaries[size_t(r)] = read;
        }
        boundaries[size_t(world)] = readCount;
        for(int r = 1; r <= world; r++) boundaries[size_t(r)] = std::max(boundaries[size_t(r)], boundaries[size_t(r) - 1]);
    }

    struct Rank {
        std::vector<uint64_t> offsets;                       // world + 1: segment of every destination in this rank's sorted output
        const void* out0 = nullptr; const void* out1 = nullptr;    // records: keys u32 / vals u64; pair keys: u64 / -
        DeviceBuffer<uint32_t> recvKeys, recvTags;
        DeviceBuffer<uint64_t> recvVals, recvPairs, recvKeys64;
        uint64_t high = 0;
        std::vector<uint64_t> usedPerIteration, histPerIteration, highPerIteration, totalPerIteration, statistics;
        std::vector<std::vector<uint32_t>> overflowPerIteration;
        std::vector<shasta_oriented_read_pair> candidates;
        uint32_t log2BucketCount = 0;
        bool onePassFits = false;
    };
    std::vector<std::unique_ptr<Rank>> rankStorage;                     // (device buffers do not move)
    for(int r = 0; r < world; r++) { rankStorage.emplace_back(new Rank()); rankStorage.back()->offsets.assign(size_t(world) + 1, 0); }
    auto rankOf = [&](int r) -> Rank& { return *rankStorage[size_t(r)]; };
    CallBarrier barrier(world);
    const bool dynamic = p.minHashIterationCount == 0;
    // A fixed number of iterations: all of them in one pass -- two exchanges and four barriers per JOB instead of per iteration
    // (lowhash0HashAll / BucketsAll / MergeAll).  SHASTA_MI355X_LOWHASH_ONE_PASS=0: iteration after iteration.
    // Whether the records of all iterations fit one sort is known to a device once its job is set up: every device says, all
    // of them must agree (lowhash0OnePassFits) -- decided below, behind the first barrier.
    const bool onePassAllowed = [&] { const char* e = std::getenv("SHASTA_MI355X_LOWHASH_ONE_PASS"); return !(e && e[0] == '0'); }()
        && !dynamic && p.minHashIterationCount <= 4096 && world <= 256;
    uint64_t highFrequencyShared = 0;
    double deviceSeconds = 0;

    try {
    onEveryDevice(world, barrier, [&](int rank) {
        Context& ctx = *contexts[size_t(rank)];
        Rank& me = rankOf(rank);
        HIP_CHECK(hipSetDevice(ctx.device));
        hipStream_t stream = ctx.stream;
        const ScopedEvent evBegin, evEnd;                   // (destroyed on every path)
        HIP_CHECK(hipEventRecord(evBegin, stream));
        try {
            lowhash0Begin(ctx, p, rank, world, boundaries.data(), &me.log2BucketCount);
            me.onePassFits = onePassAllowed && lowhash0OnePassFits(ctx);
            barrier.wait();
            bool onePass = true;
            for(int r = 0; r < world; r++) onePass = onePass && rankOf(r).onePassFits;       // (the same answer on every device)
            // This rank's segment of every rank's output, into `destination`: element size `bytes`, source pointer
            // selected by `which`.  Returns the number of elements received.
            auto pull = [&](int which, size_t bytes, void* destination) {
                uint64_t at = 0;
                if(overRccl) {
                    // The same exchange as one grouped all-to-all: this device sends every destination its segment of its own output
                    // and receives every source's segment for it, in source order (where the copies below would put them).
                    const RcclApi& api = rcclApi();
                    void* const communicator = rcclCommunicators[size_t(rank)];
                    const char* const mine = static_cast<const char*>(which == 0 ? me.out0 : me.out1);
                    // A rank whose RCCL call fails aborts EVERY communicator of the group before it throws: the other ranks, which would
                    // wait in ncclGroupEnd or in their streams for a peer that never comes, get an error from their own calls instead and
                    // throw too (the call ends with the first rank's message; the group's later calls use the peer copies and say so).
                    auto rcclCheck = [&](int code, const char* what) {
                        if(code == 0) return;
                        if(api.commAbort && !rcclAborted.exchange(true)) for(void* c : rcclCommunicators) if(c) (void)api.commAbort(c);
                        throw std::runtime_error(std::string("RCCL error in ") + what + ": " + (api.errorString ? api.errorString(code) : std::to_string(code).c_str()));
                    };
                    rcclCheck(api.groupStart(), "ncclGroupStart");
                    for(int d = 0; d < world; d++) {
                        const uint64_t begin = me.offsets[size_t(d)], count = me.offsets[size_t(d) + 1] - begin;
                        if(count) rcclCheck(api.send(mine + begin * bytes, size_t(count * bytes), RCCL_UINT8, d, communicator, stream), "ncclSend");
                    }
                    for(int s = 0; s < world; s++) {
                        const Rank& source = rankOf(s);
                        const uint64_t count = source.offsets[size_t(rank) + 1] - source.offsets[size_t(rank)];
                        if(count) rcclCheck(api.recv(static_cast<char*>(destination) + at * bytes, size_t(count * bytes), RCCL_UINT8, s, communicator, stream), "ncclRecv");
                        at += count;
                    }
                    rcclCheck(api.groupEnd(), "ncclGroupEnd");
                    return at;
                }
                for(int s = 0; s < world; s++) {
                    const Rank& source = rankOf(s);
                    const uint64_t begin = source.offsets[size_t(rank)], count = source.offsets[size_t(rank) + 1] - begin;
                    if(count == 0) continue;
                    const char* from = static_cast<const char*>(which == 0 ? source.out0 : source.out1) + begin * bytes;
                    HIP_CHECK(hipMemcpyPeerAsync(static_cast<char*>(destination) + at * bytes, ctx.device, from, contexts[size_t(s)]->device, count * bytes, stream));
                    at += count;
                }
                return at;
            };
            auto incoming = [&]() {
                uint64_t n = 0;
                for(int s = 0; s < world; s++) n += rankOf(s).offsets[size_t(rank) + 1] - rankOf(s).offsets[size_t(rank)];
                return n;
            };
            uint64_t highFrequency = 0;
            if(onePass) {
                const uint64_t I = p.minHashIterationCount;
                const uint64_t* keys = nullptr; const uint64_t* vals = nullptr;
                lowhash0HashAll(ctx, me.offsets.data(), &keys, &vals);
                me.out0 = keys; me.out1 = vals;
                barrier.wait();
                // C1: the records (of every iteration) of the buckets this device owns.
                const uint64_t records = incoming();
                me.recvKeys64.reserve(records + 1, stream); me.recvVals.reserve(records + 1, stream);
                (void)pull(0, sizeof(uint64_t), me.recvKeys64.data());
                (void)pull(1, sizeof(uint64_t), me.recvVals.data());
                HIP_CHECK(hipStreamSynchronize(stream));
                barrier.wait();                                   // every device has what it needs: the sources may be reused
                const uint64_t* pairKeys = nullptr; const uint32_t* pairTags = nullptr;
                std::vector<uint64_t> overflow;
                me.usedPerIteration.assign(size_t(I), 0);
                me.histPerIteration.assign(size_t(I) * size_t(LOWHASH0_SIZE_HISTOGRAM_BINS), 0);
                lowhash0BucketsAll(ctx, me.recvKeys64.data(), me.recvVals.data(), records, me.offsets.data(), &pairKeys, &pairTags,
                    me.usedPerIteration.data(), me.histPerIteration.data(), overflow);
                me.overflowPerIteration.assign(size_t(I), std::vector<uint32_t>());
                for(uint64_t e : overflow) me.overflowPerIteration[size_t(e >> 32)].push_back(uint32_t(e));
                me.out0 = pairKeys; me.out1 = pairTags;
                barrier.wait();
                // C2: the pair keys (with their iteration tags) whose readId0 this device owns.
                const uint64_t pairs = incoming();
                me.recvPairs.reserve(pairs + 1, stream); me.recvTags.reserve(pairs + 1, stream);
                (void)pull(0, sizeof(uint64_t), me.recvPairs.data());
                (void)pull(1, sizeof(uint32_t), me.recvTags.data());
                HIP_CHECK(hipStreamSynchronize(stream));
                barrier.wait();
                lowhash0MergeAll(ctx, me.recvPairs.data(), me.recvTags.data(), pairs);
            }
            else for(uint64_t iteration = 0; ; iteration++) {
                // Iteration control, src/LowHash0.cpp:136-157 (on the global counter: every device decides alike).
                if(dynamic) {
                    if(2. * double(highFrequency) / double(readCount) >= p.alignmentCandidatesPerRead) break;
                } else if(iteration == p.minHashIterationCount) {
                    break;
                }
                const uint32_t* keys = nullptr; const uint64_t* vals = nullptr;
                lowhash0Hash(ctx, iteration, me.offsets.data(), &keys, &vals);
                me.out0 = keys; me.out1 = vals;
                barrier.wait();
                // C1: the records of the buckets this device owns.
                const uint64_t records = incoming();
                me.recvKeys.reserve(records + 1, stream); me.recvVals.reserve(records + 1, stream);
                (void)pull(0, sizeof(uint32_t), me.recvKeys.data());
                (void)pull(1, sizeof(uint64_t), me.recvVals.data());
                HIP_CHECK(hipStreamSynchronize(stream));
                barrier.wait();                                   // every device has what it needs: the sources may be reused
                const uint64_t* pairKeys = nullptr;
                uint64_t used = 0;
                std::vector<uint64_t> hist;
                hist.resize(size_t(LOWHASH0_SIZE_HISTOGRAM_BINS));
                std::vector<uint32_t> overflow;
                lowhash0Buckets(ctx, me.recvKeys.data(), me.recvVals.data(), records, me.offsets.data(), &pairKeys, &used, hist.data(), overflow);
                me.out0 = pairKeys; me.out1 = nullptr;
                me.usedPerIteration.push_back(used);
                me.histPerIteration.insert(me.histPerIteration.end(), hist.begin(), hist.end());
                me.overflowPerIteration.push_back(overflow);
                barrier.wait();
                // C2: the pair keys whose readId0 this device owns.
                const uint64_t pairs = incoming();
                me.recvPairs.reserve(pairs + 1, stream);
                (void)pull(0, sizeof(uint64_t), me.recvPairs.data());
                HIP_CHECK(hipStreamSynchronize(stream));
                barrier.wait();
                uint64_t total = 0;
                lowhash0Merge(ctx, me.recvPairs.data(), pairs, dynamic, &me.high, &total);
                if(dynamic) {
                    barrier.wait();
                    if(rank == 0) { highFrequencyShared = 0; for(int r = 0; r < world; r++) highFrequencyShared += rankOf(r).high; }
                    barrier.wait();
                    highFrequency = highFrequencyShared;
                }
            }
            me.statistics.assign(3 * readCount, 0);
            lowhash0Finish(ctx, me.statistics.data(), me.candidates, me.highPerIteration, me.totalPerIteration);
        } catch(...) {
            ctx.lowhashJob.reset();
            throw;
        }
        HIP_CHECK(hipEventRecord(evEnd, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        if(rank == 0) { float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, evBegin, evEnd)); deviceSeconds = ms * 1e-3; }
    });
    } catch(...) {
        if(rcclAborted.load()) rcclCommunicators.clear();        // (aborted communicators are gone: the destructor must not destroy them, later calls use the peer copies)
        throw;
    }

    // Reductions and assembly, in rank order (each device's candidates are sorted and cover its readId0 range:
    // their concatenation is the reference's order).
    const uint64_t iterations = rankOf(0).highPerIteration.size();
    std::vector<shasta_oriented_read_pair> candidates;
    std::vector<uint64_t> high(iterations, 0), total(iterations, 0), histogramRows;
    std::memset(readLowHashStatistics, 0, 3 * readCount * sizeof(uint64_t));
    for(int k = 0; k < world; k++) {
        const Rank& r = rankOf(k);
        candidates.insert(candidates.end(), r.candidates.begin(), r.candidates.end());
        for(uint64_t q = 0; q < 3 * readCount; q++) readLowHashStatistics[q] += r.statistics[q];
        for(uint64_t t = 0; t < iterations; t++) { high[t] += r.highPerIteration[t]; total[t] += r.totalPerIteration[t]; }
    }
    const uint64_t bucketCount = 1ULL << rankOf(0).log2BucketCount;
    for(uint64_t t = 0; t < iterations; t++) {
        // Histogram rows (src/LowHash0.cpp:586-595) of the iteration from the summed bins.
        std::map<uint64_t, uint64_t> rows;
        uint64_t used = 0;
        for(int k = 0; k < world; k++) {
            const Rank& r = rankOf(k);
            used += r.usedPerIteration[t];
            for(int s = 1; s < LOWHASH0_SIZE_HISTOGRAM_BINS; s++) {
                const uint64_t c = r.histPerIteration[t * LOWHASH0_SIZE_HISTOGRAM_BINS + uint64_t(s)];
                if(c) rows[uint64_t(s)] += c;
            }
            for(uint32_t s : r.overflowPerIteration[t]) ++rows[s];
        }
        if(bucketCount > used) rows[0] = bucketCount - used;
        for(const auto& row : rows) { histogramRows.push_back(t); histogramRows.push_back(row.first); histogramRows.push_back(row.second); }
    }
    result.log2BucketCount = rankOf(0).log2BucketCount;
    result.candidateCount = candidates.size();
    result.candidates = mallocCopy(candidates);
    result.iterationCount = uint32_t(iterations);
    result.highFrequency = mallocCopy(high);
    result.total = mallocCopy(total);
    result.histogramRowCount = histogramRows.size() / 3;
    result.histogram = mallocCopy(histogramRows);
    result.deviceSeconds = deviceSeconds;
    result.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// computeAlignments (method 4 or 3) over the group: contiguous candidate ranges balanced by the markers they touch.
void Group::alignRun(uint64_t candidateCount, const shasta_oriented_read_pair* candidates,
    const shasta_align4_options* options4, const shasta_align3_options* options3, bool wantOrdinals, shasta_align4_result& result, bool borrowed)
{
    std::memset(&result, 0, sizeof(result));
    const auto t0 = std::chrono::steady_clock::now();
    const int world = int(contexts.size());
    // (borrowed: every device's share stays in its context's arrays; with several devices the shares are concatenated into the
    // group's own, which keep their size from call to call)
    auto runOn = [&](Context& ctx, uint64_t begin, uint64_t end, shasta_align4_result& r) {
        if(options4) align4Run(ctx, end - begin, candidates + begin, *options4, wantOrdinals, r, borrowed);
        else align3Run(ctx, end - begin, candidates + begin, *options3, wantOrdinals, r, borrowed);
    };
    if(world == 1) { runOn(*contexts[0], 0, candidateCount, result); return; }
    const uint64_t readCount = contexts[0]->readCount;
    const std::vector<uint64_t>& toc = contexts[0]->hostToc;
    // Split points: equal shares of sum(nx + ny).
    std::vector<uint64_t> cut(size_t(world) + 1, 0);
    {
        std::vector<uint64_t> prefix(candidateCount + 1, 0);
        for(uint64_t k = 0; k < candidateCount; k++) {
            const shasta_oriented_read_pair& c = candidates[k];
            if(!(c.readIds[0] < c.readIds[1]) || c.readIds[1] >= readCount) throw std::runtime_error("Align4: invalid alignment candidate (need readId0 < readId1 < readCount).");
            const uint64_t o0 = 2ULL * c.readIds[0], o1 = 2ULL * c.readIds[1] + (c.isSameStrand ? 0 : 1);
            prefix[k + 1] = prefix[k] + (toc[o0 + 1] - toc[o0]) + (toc[o1 + 1] - toc[o1]) + 64;       // + a constant per candidate: empty reads still cost a slot
        }
        for(int r = 1; r < world; r++) {
            const uint64_t target = uint64_t(double(prefix[candidateCount]) * double(r) / double(world));
            cut[size_t(r)] = uint64_t(std::lower_bound(prefix.begin(), prefix.end(), target) - prefix.begin());
            cut[size_t(r)] = std::min(std::max(cut[size_t(r)], cut[size_t(r) - 1]), candidateCount);
        }
        cut[size_t(world)] = candidateCount;
    }
    std::vector<shasta_align4_result> parts;
    parts.resize(size_t(world));
    for(shasta_align4_result& r : parts) std::memset(&r, 0, sizeof(r));
    CallBarrier barrier(world);
    try {
        onEveryDevice(world, barrier, [&](int rank) {
            HIP_CHECK(hipSetDevice(contexts[size_t(rank)]->device));
            runOn(*contexts[size_t(rank)], cut[size_t(rank)], cut[size_t(rank) + 1], parts[size_t(rank)]);
        });
    } catch(...) {
        for(shasta_align4_result& r : parts) align4Free(r);
        throw;
    }
    // Concatenation in candidate order.
    uint64_t rows = 0, bytes = 0, ordinals = 0;
    for(int r = 0; r < world; r++) {
        const shasta_align4_result& q = parts[size_t(r)];
        rows += q.alignmentCount; bytes += q.compressedToc ? q.compressedToc[q.alignmentCount] : 0;
        if(wantOrdinals && q.ordinalsToc) ordinals += q.ordinalsToc[cut[size_t(r) + 1] - cut[size_t(r)]];
    }
    auto allocate = [](size_t n) { void* p = std::malloc(std::max<size_t>(1, n)); if(!p) throw std::bad_alloc(); return p; };
    if(borrowed) {
        auto room = [](auto& v, size_t n) { if(v.size() < std::max<size_t>(1, n)) v.resize(std::max<size_t>(1, n) + n / 8); return v.data(); };
        result.alignmentData = room(storeRows, rows);
        result.compressedToc = room(storeToc, rows + 1);
        result.compressedData = room(storeBytes, bytes);
        result.status = room(storeStatus, candidateCount);
        if(wantOrdinals) { result.ordinalsToc = room(storeOrdinalsToc, candidateCount + 1); result.ordinals = room(storeOrdinals, 2 * ordinals); result.ordinalsToc[0] = 0; }
        result.owner = this;
    } else {
    result.alignmentData = static_cast<shasta_alignment_data*>(allocate(rows * sizeof(shasta_alignment_data)));
    result.compressedToc = static_cast<uint64_t*>(allocate((rows + 1) * sizeof(uint64_t)));
    result.compressedData = static_cast<uint8_t*>(allocate(bytes));
    result.status = static_cast<uint8_t*>(allocate(candidateCount));
    if(wantOrdinals) {
        result.ordinalsToc = static_cast<uint64_t*>(allocate((candidateCount + 1) * sizeof(uint64_t)));
        result.ordinals = static_cast<uint32_t*>(allocate(2 * ordinals * sizeof(uint32_t)));
        result.ordinalsToc[0] = 0;
    }
    }
    result.compressedToc[0] = 0;
    uint64_t rowBase = 0, byteBase = 0, ordBase = 0;
    for(int r = 0; r < world; r++) {
        shasta_align4_result& q = parts[size_t(r)];
        const uint64_t n = cut[size_t(r) + 1] - cut[size_t(r)];
        const uint64_t qBytes = q.compressedToc ? q.compressedToc[q.alignmentCount] : 0;
        if(q.alignmentCount) std::memcpy(result.alignmentData + rowBase, q.alignmentData, q.alignmentCount * sizeof(shasta_alignment_data));
        if(qBytes) std::memcpy(result.compressedData + byteBase, q.compressedData, qBytes);
        for(uint64_t k = 1; k <= q.alignmentCount; k++) result.compressedToc[rowBase + k] = byteBase + q.compressedToc[k];
        if(n) std::memcpy(result.status + cut[size_t(r)], q.status, n);
        if(wantOrdinals && q.ordinalsToc) {
            const uint64_t qOrdinals = q.ordinalsToc[n];
            if(qOrdinals) std::memcpy(result.ordinals + 2 * ordBase, q.ordinals, 2 * qOrdinals * sizeof(uint32_t));
            for(uint64_t k = 1; k <= n; k++) result.ordinalsToc[cut[size_t(r)] + k] = ordBase + q.ordinalsToc[k];
            ordBase += qOrdinals;
        }
        rowBase += q.alignmentCount; byteBase += qBytes;
        result.dpCellCount += q.dpCellCount; result.kmerIdBytes += q.kmerIdBytes; result.alignedBytes += q.alignedBytes;
        result.deviceSeconds = std::max(result.deviceSeconds, q.deviceSeconds);
        align4Free(q);
    }
    result.alignmentCount = rows;
    result.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace shasta_mi355x
