// Shared host-side plumbing for libshasta_mi355x.so: error propagation, device
// buffers, the context object.  gfx950 only; no CPU fallback anywhere.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace shasta_mi355x {

inline void hipCheck(hipError_t e, const char* what, const char* file, int line)
{
    if(e != hipSuccess) {
        throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " in " + what +
            " at " + file + ":" + std::to_string(line));
    }
}
#define HIP_CHECK(x) ::shasta_mi355x::hipCheck((x), #x, __FILE__, __LINE__)

// Mirrors SHASTA_ASSERT (src/SHASTA_ASSERT.hpp): throws std::runtime_error.
#define MI355X_ASSERT(x) do { if(!(x)) throw std::runtime_error( \
    std::string("Assertion failed: ") + #x + " at " + __FILE__ + ":" + std::to_string(__LINE__)); } while(0)

// double -> uint64_t exactly as the reference binary does it.  The reference is built with gcc for
// x86-64 (staticLibrary/CMakeLists.txt), where an out-of-range conversion is not an error but a
// definite bit pattern: cvttsd2si of (x - 2^63) xor 2^63 for x >= 2^63, cvttsd2si of x below, and
// cvttsd2si itself returns 0x8000000000000000 for anything outside int64 (NaN included).  The C++
// cast is undefined there and ROCm clang answers differently (hashFraction = 1.0: gcc 0, clang
// 2^63), so src/LowHash0.cpp:74 and :109 are restated as this explicit function, not as a cast.
inline uint64_t referenceDoubleToUint64(double x)
{
    const double two63 = 9223372036854775808.0;
    auto cvttsd2si = [two63](double y) -> uint64_t {
        if(!(y >= -two63 && y < two63)) return 0x8000000000000000ULL;
        return uint64_t(int64_t(y));
    };
    if(x >= two63) return cvttsd2si(x - two63) ^ 0x8000000000000000ULL;
    return cvttsd2si(x);
}

// Buffers that play the same role in the scratch of several host workers share a high-water mark: what one worker had to
// grow to, the others grow to at the start of their next batch (raiseToMark) instead of each finding out in the middle of one.
// hipFree waits for the whole device, so a reallocation stalls every worker's stream; with the marks the reallocations of a
// repeated workload end after its first pass.  A buffer is bound to the mark of its position when it is constructed while
// a SharedCapacityBinding is active on the thread (makeWorkerScratch in align4.hip); otherwise it has none.
// (api.hip) `bytes` at p filled with pseudo-random words that differ from call to call: SHASTA_MI355X_SCRAMBLE.
void scrambleDeviceMemory(void* p, size_t bytes, hipStream_t stream);
class SharedCapacityMember {
public:
    virtual void raiseToMark(hipStream_t stream) = 0;
    virtual void scramble(hipStream_t) {}
protected:
    ~SharedCapacityMember() = default;
};
struct SharedCapacities {
    static constexpr int MAX_BUFFERS = 128;
    std::atomic<size_t> marks[MAX_BUFFERS];
    SharedCapacities() { for(auto& m : marks) m.store(0); }
};
struct SharedCapacityBinding { SharedCapacities* shared; std::vector<SharedCapacityMember*>* members; int next; };
inline thread_local SharedCapacityBinding* sharedCapacityBinding = nullptr;
inline std::atomic<size_t>* bindSharedCapacity(SharedCapacityMember* member)
{
    SharedCapacityBinding* s = sharedCapacityBinding;
    if(!s) return nullptr;
    if(s->next >= SharedCapacities::MAX_BUFFERS) throw std::runtime_error("SharedCapacities: too many buffers in one scratch.");
    s->members->push_back(member);
    return &s->shared->marks[s->next++];
}
// The capacity to allocate when `wanted` is needed: at least the mark, and the mark raised to it.
inline size_t sharedCapacityFor(std::atomic<size_t>* mark, size_t wanted)
{
    if(!mark) return wanted;
    size_t m = mark->load();
    while(m < wanted && !mark->compare_exchange_weak(m, wanted)) {}
    return std::max(m, wanted);
}

// A device allocation that only grows.  HBM is 288 GB: buffers are sized once
// from the marker count and kept for the life of the context.
template<class T> class DeviceBuffer : public SharedCapacityMember {
public:
    DeviceBuffer() : mark(bindSharedCapacity(this)) {}
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer() { release(); }
    void release() { if(p) { (void)hipFree(p); p = nullptr; cap = 0; } }
    // Ensures capacity >= n.  Contents are NOT preserved unless keep is true.
    void reserve(size_t n, hipStream_t stream = nullptr, bool keep = false)
    {
        if(n <= cap) return;
        reallocate(sharedCapacityFor(mark, n + n / 8 + 64), stream, keep);
    }
    // Contents are not preserved.
    void raiseToMark(hipStream_t stream) override
    {
        const size_t m = mark ? mark->load() : 0;
        if(m > cap) reallocate(m, stream, false);
    }
    // SHASTA_MI355X_SCRAMBLE=1 (a test switch): the whole buffer overwritten with pseudo-random data at the head of every batch -- what a kernel
    // reads without its batch having written it is then neither what an earlier, similar batch left there nor a constant byte.
    void scramble(hipStream_t stream) override { if(p && cap) scrambleDeviceMemory(p, cap * sizeof(T), stream); }
    T* data() const { return p; }
    size_t capacity() const { return cap; }
    void swap(DeviceBuffer& o) { std::swap(p, o.p); std::swap(cap, o.cap); }
private:
    void reallocate(size_t newCap, hipStream_t stream, bool keep)
    {
        T* q = nullptr;
        // SHASTA_MI355X_LOG_ALLOC=1: every (re)allocation of a device buffer on stderr -- a steady state allocates nothing.
        static const bool logAllocations = [] { const char* e = std::getenv("SHASTA_MI355X_LOG_ALLOC"); return e && e[0] == '1'; }();
        if(logAllocations) std::fprintf(stderr, "shasta_mi355x: device buffer %zu -> %zu bytes at %.1f ms\n", cap * sizeof(T), newCap * sizeof(T),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count());      // (the clock of python's time.monotonic())
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&q), newCap * sizeof(T)));
        // SHASTA_MI355X_POISON=<byte>: every new device buffer filled with that byte (a test switch: a kernel that reads memory nothing
        // wrote gives results that change with the value; fresh hipMalloc memory otherwise holds whatever the process freed before).
        static const int poison = [] { const char* e = std::getenv("SHASTA_MI355X_POISON"); return e ? int(std::strtol(e, nullptr, 0)) & 0xff : -1; }();
        if(poison >= 0) { HIP_CHECK(hipMemsetAsync(q, poison, newCap * sizeof(T), stream)); HIP_CHECK(hipStreamSynchronize(stream)); HIP_CHECK(hipDeviceSynchronize()); }
        // SHASTA_MI355X_SCRAMBLE=1: ... and with pseudo-random data that differs from allocation to allocation (the aligner's workers also
        // scramble their scratch at the head of every batch: align4.hip).
        static const bool scrambleNew = [] { const char* e = std::getenv("SHASTA_MI355X_SCRAMBLE"); return e && e[0] == '1'; }();
        if(scrambleNew) { scrambleDeviceMemory(q, newCap * sizeof(T), stream); HIP_CHECK(hipStreamSynchronize(stream)); }
        if(keep && p && cap) {
            HIP_CHECK(hipMemcpyAsync(q, p, cap * sizeof(T), hipMemcpyDeviceToDevice, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
        }
        if(p) (void)hipFree(p);
        p = q; cap = newCap;
    }
    T* p = nullptr;
    size_t cap = 0;
    std::atomic<size_t>* mark;
};

// Page-locked host staging memory that only grows (device-to-host copies into pageable memory go
// through the runtime's small bounce buffers; into pinned memory they run at PCIe speed).
class PinnedBuffer : public SharedCapacityMember {
public:
    PinnedBuffer() : mark(bindSharedCapacity(this)) {}
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    ~PinnedBuffer() { if(p) (void)hipHostFree(p); }
    void* reserve(size_t bytes)
    {
        if(bytes > cap) reallocate(sharedCapacityFor(mark, bytes + bytes / 4 + 4096));
        return p;
    }
    void raiseToMark(hipStream_t) override
    {
        const size_t m = mark ? mark->load() : 0;
        if(m > cap) reallocate(m);
    }
    void* data() const { return p; }
private:
    void reallocate(size_t newCap)
    {
        if(p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        HIP_CHECK(hipHostMalloc(&p, newCap, hipHostMallocDefault));
        cap = newCap;
    }
    void* p = nullptr;
    size_t cap = 0;
    std::atomic<size_t>* mark;
};

// A stream / an event that is destroyed on every path out of its scope.
struct ScopedStream {
    hipStream_t stream = nullptr;
    ScopedStream() { HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); }
    ScopedStream(const ScopedStream&) = delete;
    ScopedStream& operator=(const ScopedStream&) = delete;
    ~ScopedStream() { if(stream) (void)hipStreamDestroy(stream); }
    operator hipStream_t() const { return stream; }
};
struct ScopedEvent {
    hipEvent_t event = nullptr;
    ScopedEvent() { HIP_CHECK(hipEventCreate(&event)); }
    ScopedEvent(const ScopedEvent&) = delete;
    ScopedEvent& operator=(const ScopedEvent&) = delete;
    ~ScopedEvent() { if(event) (void)hipEventDestroy(event); }
    operator hipEvent_t() const { return event; }
};

struct EventTimer {
    hipEvent_t a = nullptr, b = nullptr;
    EventTimer() { HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b)); }
    ~EventTimer() { if(a) (void)hipEventDestroy(a); if(b) (void)hipEventDestroy(b); }
    void start(hipStream_t s) { HIP_CHECK(hipEventRecord(a, s)); }
    void stop(hipStream_t s) { HIP_CHECK(hipEventRecord(b, s)); }
    double seconds() { HIP_CHECK(hipEventSynchronize(b)); float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b)); return ms * 1e-3; }
};

// An invariant of device code that the emulated build (tests/emu, -DSHASTA_DEVICE_CHECK=...) verifies on
// every lane; compiled out of the product.
#ifndef SHASTA_DEVICE_CHECK
#define SHASTA_DEVICE_CHECK(x) ((void)0)
#endif

inline unsigned divUp(uint64_t a, uint64_t b) { return unsigned((a + b - 1) / b); }

// Per-kernel timing table of a context (shasta_mi355x_kernel_table): every launch of the stages is bracketed by two
// HIP events recorded on the stream it is issued on; the durations, launch counts, algorithmic bytes (SURVEY 8d's
// per-unit figure x the units of the launch) and a kernel-specific work count accumulate per kernel name until the
// table is reset.  The two host workers of the aligner use it concurrently.
class KernelTimers {
public:
    struct Entry { std::string name; double seconds = 0; uint64_t launches = 0, bytes = 0, work = 0; };
    struct Span { int id = -1; hipEvent_t begin = nullptr, end = nullptr; hipStream_t stream = nullptr; };
    ~KernelTimers() { for(hipEvent_t e : pool) (void)hipEventDestroy(e); for(Pending& p : pending) { (void)hipEventDestroy(p.begin); (void)hipEventDestroy(p.end); } }
    // SHASTA_MI355X_KERNEL_TIMERS=0: no events around the launches (a caller that never reads the table; spans are no-ops).
    KernelTimers() { const char* e = std::getenv("SHASTA_MI355X_KERNEL_TIMERS"); enabled = !(e && e[0] == '0'); }
    Span begin(const char* name, hipStream_t stream)
    {
        Span s;
        s.stream = stream;
        if(!enabled) return s;
        {
            std::lock_guard<std::mutex> lock(mutex);
            s.id = idOf(name);
            s.begin = take(); s.end = take();
        }
        HIP_CHECK(hipEventRecord(s.begin, stream));
        return s;
    }
    // Returns a handle with which bytes / work can be set later (counts that are only known after a read-back).
    size_t end(const Span& s, uint64_t bytes = 0, uint64_t work = 0)
    {
        if(s.id < 0) return ~size_t(0);
        HIP_CHECK(hipEventRecord(s.end, s.stream));
        std::lock_guard<std::mutex> lock(mutex);
        if(pending.size() >= 8192) collectLocked();       // a caller that never reads the table must not pile up events
        pending.push_back(Pending{s.id, s.begin, s.end, bytes, work, nextHandle});
        return nextHandle++;
    }
    // (Launches of several streams finish out of order and are folded as they finish: a handle names its launch, not a
    // position in the list; a launch that has been folded already keeps what it was booked with.)
    void amend(size_t handle, uint64_t bytes, uint64_t work)
    {
        std::lock_guard<std::mutex> lock(mutex);
        for(Pending& p : pending) if(p.handle == handle) { p.bytes = bytes; p.work = work; return; }
    }
    // A row that counts something other than launches (no time): what the caller says, added up.
    void count(const char* name, uint64_t launches, uint64_t bytes, uint64_t work)
    {
        if(!enabled) return;
        std::lock_guard<std::mutex> lock(mutex);
        Entry& e = entries[size_t(idOf(name))];
        e.launches += launches; e.bytes += bytes; e.work += work;
    }
    // Folds every finished launch into the table.  The caller has synchronised the streams it launched on;
    // a launch that is still running stays pending.
    void collect() { std::lock_guard<std::mutex> lock(mutex); collectLocked(); }
    void reset() { std::lock_guard<std::mutex> lock(mutex); collectLocked(); for(Entry& e : entries) { e.seconds = 0; e.launches = e.bytes = e.work = 0; } }
    std::vector<Entry> table() { std::lock_guard<std::mutex> lock(mutex); collectLocked(); return entries; }
private:
    void collectLocked()
    {
        size_t kept = 0;
        for(size_t k = 0; k < pending.size(); k++) {
            Pending& p = pending[k];
            float ms = 0;
            if(hipEventQuery(p.end) == hipSuccess && hipEventElapsedTime(&ms, p.begin, p.end) == hipSuccess) {
                Entry& e = entries[size_t(p.id)];
                e.seconds += double(ms) * 1e-3; e.launches += 1; e.bytes += p.bytes; e.work += p.work;
                pool.push_back(p.begin); pool.push_back(p.end);
            } else {
                pending[kept++] = p;
            }
        }
        pending.resize(kept);
    }
    struct Pending { int id; hipEvent_t begin, end; uint64_t bytes, work; size_t handle; };
    int idOf(const char* name)
    {
        for(size_t k = 0; k < entries.size(); k++) if(entries[k].name == name) return int(k);
        entries.push_back(Entry()); entries.back().name = name;
        return int(entries.size() - 1);
    }
    hipEvent_t take()
    {
        if(!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        return e;
    }
    std::mutex mutex;
    std::vector<Entry> entries;
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    size_t nextHandle = 0;
    bool enabled = true;
};

}  // namespace shasta_mi355x
