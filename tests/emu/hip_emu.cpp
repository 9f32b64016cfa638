// TEST INFRASTRUCTURE ONLY -- see include/hip/hip_runtime.h.
// Fiber scheduler (one OS thread runs one workgroup at a time; its work-items are fibers that
// switch only at cross-lane operations and barriers) and the host runtime stubs.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <elf.h>
#include <link.h>
#include <sys/mman.h>

#include <atomic>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <functional>
#include <map>
#include <cstdio>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

// void hipemu_switch(void** saveSp, void* loadSp): System V x86-64 callee-saved registers.
extern "C" void hipemu_switch(void** saveSp, void* loadSp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

thread_local Fiber* cur = nullptr;

namespace {

enum State { RUNNABLE = 0, WAIT_WAVE, WAIT_BLOCK, DONE };
constexpr size_t STACK_BYTES = 256 * 1024;

struct Worker {                       // per OS thread
    void* schedulerSp = nullptr;
    char* stacks = nullptr;
    size_t stackCount = 0;
    const Launch* launch = nullptr;
    std::string error;
    ~Worker() { if(stacks) munmap(stacks, stackCount * STACK_BYTES); }
    void reserve(size_t n)
    {
        if(n <= stackCount) return;
        if(stacks) munmap(stacks, stackCount * STACK_BYTES);
        stacks = static_cast<char*>(mmap(nullptr, n * STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
        if(stacks == MAP_FAILED) { stacks = nullptr; throw std::runtime_error("hipemu: cannot map fiber stacks"); }
        stackCount = n;
    }
};
thread_local Worker worker;

void yieldToScheduler()
{
    Fiber* f = cur;
    hipemu_switch(&f->sp, worker.schedulerSp);
}

void fiberEntry()
{
    Fiber* f = cur;
    try {
        worker.launch->invoke(worker.launch->args);
    } catch(const std::exception& e) {
        worker.error = e.what();
    }
    f->state = DONE;
    yieldToScheduler();
    std::abort();                     // a finished fiber is never resumed
}

void prepare(Fiber& f, char* stackTop)
{
    // Layout, low to high: r15 r14 r13 r12 rbx rbp | return address = fiberEntry | (alignment slot).
    uintptr_t top = reinterpret_cast<uintptr_t>(stackTop) & ~uintptr_t(15);
    void** p = reinterpret_cast<void**>(top);
    *--p = nullptr;                                   // so that rsp % 16 == 8 on entry, as after a call
    *--p = reinterpret_cast<void*>(&fiberEntry);
    for(int i = 0; i < 6; i++) *--p = nullptr;
    f.sp = p;
}

// HIPEMU_LDS_SCRAMBLE=<seed>: LDS is not cleared between workgroups on the hardware -- a workgroup finds what the last one on its CU
// left there, of whatever kernel of whatever process.  Here a __shared__ variable is a function-scope thread_local: zero when the OS
// thread first meets it and the SAME kernel's residue afterwards, which forgives a read of LDS the workgroup has not written.  With the
// switch every __shared__ variable of the library is filled with pseudo-random data before every workgroup (all ones, small integers and
// small signed values take turns with random words: residue that looks like counts, indices and scores), and a shuffle whose source lane
// is switched off returns garbage instead of the reader's own value.  The variables are found in the library's own symbol table:
// the STT_TLS symbols of local statics (_ZZ...) and the arrays that stand in for the dynamic LDS (dynamic_lds_*.h).
struct LdsMap {
    bool on = false;
    uint64_t seed = 0;
    std::vector<std::pair<size_t, size_t>> ranges;     // (offset in the module's TLS block, bytes)
    size_t tlsBytes = 0, totalBytes = 0;
};
LdsMap buildLdsMap()
{
    LdsMap m;
    const char* e = std::getenv("HIPEMU_LDS_SCRAMBLE");
    if(!e || !e[0] || (e[0] == '0' && !e[1])) return m;
    m.seed = std::strtoull(e, nullptr, 0);
    Dl_info info;
    if(!dladdr(reinterpret_cast<void*>(&hipemu::launch), &info) || !info.dli_fname) { std::fprintf(stderr, "hipemu: HIPEMU_LDS_SCRAMBLE: the library's file is unknown\n"); std::abort(); }
    std::ifstream in(info.dli_fname, std::ios::binary);
    std::vector<char> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if(file.size() < sizeof(Elf64_Ehdr)) { std::fprintf(stderr, "hipemu: HIPEMU_LDS_SCRAMBLE: cannot read %s\n", info.dli_fname); std::abort(); }
    const Elf64_Ehdr* eh = reinterpret_cast<const Elf64_Ehdr*>(file.data());
    const Elf64_Phdr* ph = reinterpret_cast<const Elf64_Phdr*>(file.data() + eh->e_phoff);
    for(int i = 0; i < eh->e_phnum; i++) if(ph[i].p_type == PT_TLS) m.tlsBytes = ph[i].p_memsz;
    const Elf64_Shdr* sh = reinterpret_cast<const Elf64_Shdr*>(file.data() + eh->e_shoff);
    const Elf64_Shdr* symtab = nullptr;
    for(int i = 0; i < eh->e_shnum; i++) if(sh[i].sh_type == SHT_SYMTAB) symtab = &sh[i];
    if(!symtab || !m.tlsBytes) { std::fprintf(stderr, "hipemu: HIPEMU_LDS_SCRAMBLE: %s has no symbol table or no TLS segment\n", info.dli_fname); std::abort(); }
    const char* names = file.data() + sh[symtab->sh_link].sh_offset;
    const Elf64_Sym* syms = reinterpret_cast<const Elf64_Sym*>(file.data() + symtab->sh_offset);
    const size_t count = symtab->sh_size / sizeof(Elf64_Sym);
    for(size_t i = 0; i < count; i++) {
        if(ELF64_ST_TYPE(syms[i].st_info) != STT_TLS || syms[i].st_size == 0) continue;
        const std::string name = names + syms[i].st_name;
        const bool localStatic = name.compare(0, 3, "_ZZ") == 0;
        const bool dynamicLds = name.find("ldsWords") != std::string::npos || name.find("wideRows") != std::string::npos || name.find("screenWindows") != std::string::npos;
        if(!localStatic && !dynamicLds) continue;
        if(syms[i].st_value + syms[i].st_size > m.tlsBytes) { std::fprintf(stderr, "hipemu: HIPEMU_LDS_SCRAMBLE: %s lies outside the TLS segment\n", name.c_str()); std::abort(); }
        m.ranges.emplace_back(size_t(syms[i].st_value), size_t(syms[i].st_size));
    }
    std::sort(m.ranges.begin(), m.ranges.end());
    m.ranges.erase(std::unique(m.ranges.begin(), m.ranges.end()), m.ranges.end());
    for(const auto& r : m.ranges) m.totalBytes += r.second;
    std::fprintf(stderr, "hipemu: HIPEMU_LDS_SCRAMBLE: %zu __shared__ variables, %zu bytes, filled before every workgroup\n", m.ranges.size(), m.totalBytes);
    m.on = true;
    return m;
}
const LdsMap& ldsMap() { static const LdsMap m = buildLdsMap(); return m; }

thread_local char* tlsBlock = nullptr;
thread_local uint64_t garbageState = 0, scheduleState = 0;
std::atomic<uint64_t> scrambleCounter(0);

inline uint64_t nextGarbage(uint64_t& s) { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1DULL; }

char* tlsBlockOfThisThread(size_t bytes)
{
    if(tlsBlock) return tlsBlock;
    struct Query { const void* inside; char* block; } q{reinterpret_cast<const void*>(&hipemu::launch), nullptr};
    (void)cur;                                           // (the module's block exists in this thread once one of its variables was touched)
    dl_iterate_phdr([](dl_phdr_info* info, size_t, void* data) {
        Query* q = static_cast<Query*>(data);
        for(int i = 0; i < info->dlpi_phnum; i++) {
            const ElfW(Phdr)& p = info->dlpi_phdr[i];
            const uintptr_t a = info->dlpi_addr + p.p_vaddr, x = reinterpret_cast<uintptr_t>(q->inside);
            if(p.p_type == PT_LOAD && x >= a && x < a + p.p_memsz) { q->block = static_cast<char*>(info->dlpi_tls_data); return 1; }
        }
        return 0;
    }, &q);
    const char* own = reinterpret_cast<const char*>(&cur);
    if(!q.block || own < q.block || own >= q.block + bytes) { std::fprintf(stderr, "hipemu: HIPEMU_LDS_SCRAMBLE: this thread's TLS block was not found\n"); std::abort(); }
    return tlsBlock = q.block;
}

void scrambleLds()
{
    const LdsMap& m = ldsMap();
    if(!m.on) return;
    char* block = tlsBlockOfThisThread(m.tlsBytes);
    const uint64_t n = scrambleCounter.fetch_add(1);
    uint64_t s = (m.seed + 1) * 0x9E3779B97F4A7C15ULL + n * 0xD1B54A32D192ED03ULL; if(!s) s = 1;
    const unsigned mode = unsigned(n & 3u);
    for(const auto& r : m.ranges) {
        char* p = block + r.first; size_t left = r.second;
        while(left) {
            uint64_t w = nextGarbage(s);
            if(mode == 1) w = ~uint64_t(0);
            else if(mode == 2) w &= 0x00000fff00000fffULL;                                                  // small counts and indices
            else if(mode == 3) { const uint32_t a = uint32_t(int32_t(w % 6000) - 1000), b = uint32_t(int32_t((w >> 32) % 6000) - 1000); w = (uint64_t(b) << 32) | a; }   // scores
            const size_t k = std::min<size_t>(left, 8);
            std::memcpy(p, &w, k); p += k; left -= k;
        }
    }
    garbageState = s;
}

// Completes the collective of the lanes in `group` (all waiting at the same call site).
void releaseGroup(std::vector<Fiber>& fibers, int waveBase, uint64_t group)
{
    uint64_t ballot = 0;
    int first = -1;
    for(int l = 0; l < 64; l++) if((group >> l) & 1) {
        const Fiber& f = fibers[waveBase + l];
        if(first < 0) first = l;
        if(f.kind == BALLOT && f.value) ballot |= 1ULL << l;
    }
    for(int l = 0; l < 64; l++) if((group >> l) & 1) {
        Fiber& f = fibers[waveBase + l];
        switch(f.kind) {
        case BALLOT: f.result = ballot; break;
        case SHUFFLE: {
            const int src = int(f.aux);
            // An inactive source lane returns garbage on hardware; the own value keeps runs reproducible (HIPEMU_LDS_SCRAMBLE: garbage).
            f.result = (src >= 0 && src < 64 && ((group >> src) & 1)) ? fibers[waveBase + src].value : (garbageState ? nextGarbage(garbageState) : f.value);
            break;
        }
        case DPP_MOVE: {                // bit 32: the source lane is active
            const int src = int(f.aux);
            f.result = (src >= 0 && src < 64 && ((group >> src) & 1)) ? ((1ULL << 32) | uint32_t(fibers[waveBase + src].value)) : 0;
            break;
        }
        case FIRSTLANE: f.result = fibers[waveBase + first].value; break;
        default: f.result = 0; break;
        }
    }
    for(int l = 0; l < 64; l++) if((group >> l) & 1) fibers[waveBase + l].state = RUNNABLE;
}

void runBlock(const Launch& L, unsigned bx, unsigned by, unsigned bz)
{
    const unsigned n = L.block.x * L.block.y * L.block.z;
    if(L.block.y != 1 || L.block.z != 1) throw std::runtime_error("hipemu: only one-dimensional workgroups are supported");
    worker.reserve(n);
    worker.launch = &L;
    worker.error.clear();
    scrambleLds();
    std::vector<Fiber> fibers(n);
    for(unsigned t = 0; t < n; t++) {
        Fiber& f = fibers[t];
        f.tIdx = Index{t, 0, 0}; f.bIdx = Index{bx, by, bz};
        f.bDim = Index{L.block.x, L.block.y, L.block.z}; f.gDim = Index{L.grid.x, L.grid.y, L.grid.z};
        f.lane = int(t & 63); f.wave = int(t >> 6); f.state = RUNNABLE; f.kind = 0; f.site = nullptr;
        f.value = f.aux = f.result = 0;
        prepare(f, worker.stacks + size_t(t + 1) * STACK_BYTES);
    }
    const unsigned waves = (n + 63) / 64;
    // HIPEMU_SCHEDULE=<seed>: the wavefronts of a workgroup run in a random order that changes at every pass, and a wavefront whose lanes
    // all wait at one cross-lane operation is let go only every other time (by a coin): wavefronts get ahead of each other by several
    // such operations, as they do on a CU, instead of advancing side by side -- LDS traffic between wavefronts that lacks a
    // barrier then shows as a changed result.  (The lanes of a wavefront keep their order: lock-step code relies on it, see above.)
    static const uint64_t scheduleSeed = [] { const char* e = std::getenv("HIPEMU_SCHEDULE"); return e && e[0] ? std::strtoull(e, nullptr, 0) + 1 : 0; }();
    const bool shuffled = scheduleSeed != 0;
    if(shuffled && !scheduleState) scheduleState = scheduleSeed * 0x9E3779B97F4A7C15ULL + scrambleCounter.fetch_add(1) * 0xD1B54A32D192ED03ULL + 1;
    std::vector<unsigned> waveOrder(waves);
    for(unsigned w = 0; w < waves; w++) waveOrder[w] = w;
    unsigned done = 0;
    while(done < n) {
        bool progressed = false;
        // Within a wavefront the highest lane runs first, so the usual leader (lane 0) runs last: a
        // follower's LDS read that precedes the leader's LDS write in program order, with no
        // cross-lane operation in between, sees the old value -- as it does in lock-step.
        if(shuffled) for(unsigned k = waves; k > 1; k--) std::swap(waveOrder[k - 1], waveOrder[nextGarbage(scheduleState) % k]);
        for(unsigned q = 0; q < waves * 64u; q++) {
            const unsigned w = waveOrder[q >> 6], top = std::min(n, (w + 1) * 64u) - 1;
            const unsigned t = top - (q & 63u);
            if(t < w * 64u || t >= n) continue;
            Fiber& f = fibers[t];
            if(f.state != RUNNABLE) continue;
            progressed = true;
            cur = &f;
            hipemu_switch(&worker.schedulerSp, f.sp);
            cur = nullptr;
            if(f.state == DONE) ++done;
        }
        // Wavefronts whose live lanes all wait at one call site: the common case.
        bool held = false;
        for(unsigned w = 0; w < waves; w++) {
            const int base = int(w * 64), count = int(std::min(64u, n - w * 64));
            uint64_t waiting = 0, live = 0;
            const void* site = nullptr; bool same = true;
            for(int l = 0; l < count; l++) {
                const Fiber& f = fibers[base + l];
                if(f.state == DONE) continue;
                live |= 1ULL << l;
                if(f.state == WAIT_WAVE) {
                    if(!waiting) site = f.site; else if(f.site != site) same = false;
                    waiting |= 1ULL << l;
                }
            }
            if(waiting && waiting == live && same) {
                if(shuffled && (progressed || w + 1 < waves) && (nextGarbage(scheduleState) & 1u)) { held = true; continue; }      // (held back this time)
                releaseGroup(fibers, base, waiting); progressed = true;
            }
        }
        if(held && !progressed) continue;                  // (every ready wavefront was held back: toss again)
        // Workgroup barrier: every live work-item waits at it.
        {
            unsigned atBarrier = 0;
            for(unsigned t = 0; t < n; t++) if(fibers[t].state == WAIT_BLOCK) ++atBarrier;
            if(atBarrier && atBarrier == n - done) {
                for(unsigned t = 0; t < n; t++) if(fibers[t].state == WAIT_BLOCK) fibers[t].state = RUNNABLE;
                progressed = true;
            }
        }
        if(progressed) continue;
        // Divergence: the live lanes of a wavefront wait at different call sites (or some at the
        // workgroup barrier).  The hardware runs the deeper branch first with only its lanes active;
        // code layout puts that branch at the lower address.
        bool released = false;
        for(unsigned w = 0; w < waves && !released; w++) {
            const int base = int(w * 64), count = int(std::min(64u, n - w * 64));
            const void* lowest = nullptr;
            for(int l = 0; l < count; l++) {
                const Fiber& f = fibers[base + l];
                if(f.state == WAIT_WAVE && (!lowest || f.site < lowest)) lowest = f.site;
            }
            if(!lowest) continue;
            uint64_t group = 0;
            for(int l = 0; l < count; l++) if(fibers[base + l].state == WAIT_WAVE && fibers[base + l].site == lowest) group |= 1ULL << l;
            releaseGroup(fibers, base, group);
            released = true;
        }
        if(!released) throw std::runtime_error("hipemu: deadlock (work-items wait at a workgroup barrier that others never reach)");
    }
    if(!worker.error.empty()) throw std::runtime_error("hipemu: exception inside a kernel: " + worker.error);
}

std::mutex errorMutex;

}  // namespace

uint64_t collective(int kind, uint64_t value, uint64_t aux)
{
    Fiber* f = cur;
    f->kind = kind; f->value = value; f->aux = aux;
    f->site = __builtin_return_address(0);
    f->state = (kind == BLOCK_BARRIER) ? WAIT_BLOCK : WAIT_WAVE;
    yieldToScheduler();
    return f->result;
}

// Helper threads that live as long as the process: a launch hands out tickets for its block loop, runs the loop itself too,
// and when its own pass is over takes back the tickets nobody picked up and waits for the helpers that did.  (A thread per
// launch mapped and unmapped its fiber stacks every time: most of a test run was spent in the kernel's memory management.)
struct Job { const std::function<void()>* work; int active = 0; };
struct HelperPool {
    std::mutex mutex;
    std::condition_variable ticketPosted, helperDone;
    std::deque<Job*> tickets;
    unsigned threadCount = 0;
};
HelperPool& helperPool() { static HelperPool* pool = new HelperPool; return *pool; }       // (never destroyed: its threads never end)

void helperLoop()
{
    HelperPool& pool = helperPool();
    for(;;) {
        Job* job = nullptr;
        {
            std::unique_lock<std::mutex> lock(pool.mutex);
            pool.ticketPosted.wait(lock, [&] { return !pool.tickets.empty(); });
            job = pool.tickets.front(); pool.tickets.pop_front();
            ++job->active;
        }
        (*job->work)();
        {
            std::lock_guard<std::mutex> lock(pool.mutex);
            --job->active;
        }
        pool.helperDone.notify_all();
    }
}

void runWithHelpers(const std::function<void()>& work, unsigned helpers, unsigned poolSize)
{
    HelperPool& pool = helperPool();
    Job job; job.work = &work;
    {
        std::lock_guard<std::mutex> lock(pool.mutex);
        for(; pool.threadCount < poolSize; ++pool.threadCount) std::thread(helperLoop).detach();
        for(unsigned k = 0; k < helpers; k++) pool.tickets.push_back(&job);
    }
    pool.ticketPosted.notify_all();
    work();
    std::unique_lock<std::mutex> lock(pool.mutex);
    pool.tickets.erase(std::remove(pool.tickets.begin(), pool.tickets.end(), &job), pool.tickets.end());
    pool.helperDone.wait(lock, [&] { return job.active == 0; });
}

void countLaunch();
// HIPEMU_PROFILE=1: at exit, the wall-clock seconds and launches of every kernel (by the symbol of its launch stub), largest first,
// on stderr.  The emulator's time is lane-operations, not device time: what it shows is WORK -- a kernel whose emulated time is out of
// proportion to its input is doing something a device would also pay for (round 6: a full cell table walked again for every match).
namespace {
struct ProfileRow { double seconds = 0; uint64_t launches = 0, blocks = 0; const char* name = "?"; };
std::mutex profileMutex;
std::map<const void*, ProfileRow>& profileRows() { static auto* const rows = new std::map<const void*, ProfileRow>; return *rows; }      // (never destroyed: the exit handler reads it)
void profilePrint()
{
    std::vector<ProfileRow> rows;
    for(const auto& r : profileRows()) rows.push_back(r.second);
    std::sort(rows.begin(), rows.end(), [](const ProfileRow& a, const ProfileRow& b) { return a.seconds > b.seconds; });
    std::fprintf(stderr, "hipemu profile (wall-clock seconds on the host, launches, workgroups; one row per instantiation):\n");
    for(const ProfileRow& r : rows) std::fprintf(stderr, "  %9.2f s %7llu %9llu  %s\n", r.seconds, (unsigned long long)r.launches, (unsigned long long)r.blocks, r.name);
}
const bool profileOn = [] { const char* e = std::getenv("HIPEMU_PROFILE"); const bool on = e && e[0] == '1'; if(on) std::atexit(profilePrint); return on; }();
}
void launchBody(const Launch& L);
void launch(const Launch& L)
{
    if(!profileOn) { launchBody(L); return; }
    const auto t0 = std::chrono::steady_clock::now();
    launchBody(L);
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::lock_guard<std::mutex> lock(profileMutex);
    ProfileRow& r = profileRows()[L.kernel];
    r.name = L.name; r.seconds += s; r.launches += 1; r.blocks += uint64_t(L.grid.x) * L.grid.y * L.grid.z;
}
void launchBody(const Launch& L)
{
    countLaunch();
    const uint64_t blocks = uint64_t(L.grid.x) * L.grid.y * L.grid.z;
    if(blocks == 0 || L.block.x == 0) throw std::runtime_error("hipemu: empty launch configuration");
    static const unsigned maxThreads = [] {
        const char* e = std::getenv("HIPEMU_THREADS");
        const unsigned n = e ? unsigned(std::atoi(e)) : std::thread::hardware_concurrency();
        return std::max(1u, n);
    }();
    std::atomic<uint64_t> next(0);
    std::string error;
    // HIPEMU_SCHEDULE: the workgroups of a launch are handed out in a scattered order (a stride coprime to their number, another one for
    // every launch) instead of 0, 1, 2 ...: a kernel whose workgroup k relies on workgroup k - 1 having run would show.
    static const bool scattered = [] { const char* e = std::getenv("HIPEMU_SCHEDULE"); return e && e[0]; }();
    uint64_t stride = 1, offset = 0;
    if(scattered && blocks > 2) {
        uint64_t s = scrambleCounter.fetch_add(1) * 0x9E3779B97F4A7C15ULL + 12345;
        offset = nextGarbage(s) % blocks;
        stride = 1 + nextGarbage(s) % (blocks - 1);
        while(std::__gcd(stride, blocks) != 1) stride = stride % (blocks - 1) + 1;
    }
    const std::function<void()> work = [&]() {
        try {
            for(;;) {
                const uint64_t ticket = next.fetch_add(1);
                if(ticket >= blocks) break;
                const uint64_t b = uint64_t((static_cast<unsigned __int128>(ticket) * stride + offset) % blocks);
                const unsigned bx = unsigned(b % L.grid.x), by = unsigned((b / L.grid.x) % L.grid.y), bz = unsigned(b / (uint64_t(L.grid.x) * L.grid.y));
                runBlock(L, bx, by, bz);
            }
        } catch(const std::exception& e) {
            std::lock_guard<std::mutex> lock(errorMutex);
            if(error.empty()) error = e.what();
            next.store(blocks);
        }
    };
    const unsigned threads = unsigned(std::min<uint64_t>(maxThreads, blocks));
    if(threads <= 1) work();
    else runWithHelpers(work, threads - 1, maxThreads - 1);
    if(!error.empty()) throw std::runtime_error(error);
}

}  // namespace hipemu

// ---------------------------------------------------------------------------
// Host runtime.
// ---------------------------------------------------------------------------
// What the host asked of the runtime so far, by kind (hipemu_api_counts: tests and scripts/emu_api_counts.py read the difference
// between two points of a run -- synchronisations, copies, launches and allocations per step are the same here as on the device).
namespace { enum { N_LAUNCH, N_STREAM_SYNC, N_DEVICE_SYNC, N_EVENT_SYNC, N_COPY_H2D, N_COPY_D2H, N_COPY_D2D, N_MEMSET, N_MALLOC, N_FREE, N_HOST_MALLOC, N_HOST_FREE, N_EVENT_RECORD, N_STREAM_WAIT, N_EVENT_CREATE, N_STREAM_CREATE, N_COPY_D2H_PAGEABLE, N_KINDS };
std::atomic<uint64_t> apiCounts[N_KINDS]; inline void counted(int kind) { apiCounts[kind].fetch_add(1, std::memory_order_relaxed); }
inline void countedCopy(hipMemcpyKind k) { counted(k == hipMemcpyHostToDevice ? N_COPY_H2D : (k == hipMemcpyDeviceToHost ? N_COPY_D2H : N_COPY_D2D)); } }
namespace hipemu { void countLaunch() { counted(N_LAUNCH); } }
extern "C" int hipemu_api_counts(uint64_t* out, int room)
{
    for(int k = 0; k < N_KINDS && k < room; k++) out[k] = apiCounts[k].load();
    return N_KINDS;
}
extern "C" const char* hipemu_api_count_names() { return "launch stream_sync device_sync event_sync copy_h2d copy_d2h copy_d2d memset malloc free host_malloc host_free event_record stream_wait_event event_create stream_create copy_d2h_to_pageable_memory"; }

// Streams and events.  Default: every launch and copy completes before its call returns (streams and events order nothing because
// nothing is ever pending).  HIPEMU_ASYNC=<seed>: the work WAITS in its stream's queue, as it does on the device, and runs as late as
// the host's own synchronisation allows -- when the stream, an event recorded behind it or the device is synchronised, when something
// that waits for one of its events has to run, when memory is freed -- with the other streams' queues drained up to a random point
// first.  A kernel that reads what another stream has not been told to finish, a host read of a result before the synchronisation
// that delivers it, a pinned source overwritten while its copy is pending: each answers differently from the oracle here, where the
// immediate mode (and, most of the time, the device) forgives it.  Copies from pageable host memory are staged at the call and
// copies to pageable host memory complete before the call returns, as the runtime does it; pinned memory is read and written
// when the copy runs.
struct hipemuEvent {
    std::chrono::steady_clock::time_point t;
    uint64_t recorded = 0, completed = 0;       // record calls so far; the last one that has run
    hipemuStream* stream = nullptr;             // where the last record waits
};
namespace {
struct Op {
    std::function<void()> run;
    hipemuEvent* records = nullptr; uint64_t recordId = 0;
    hipemuEvent* waitsFor = nullptr; uint64_t waitId = 0;
};
}
struct hipemuStream { std::deque<Op> queue; };
namespace {
const uint64_t asyncSeed = [] { const char* e = std::getenv("HIPEMU_ASYNC"); return e && e[0] && !(e[0] == '0' && !e[1]) ? std::strtoull(e, nullptr, 0) + 1 : 0; }();
std::recursive_mutex deviceMutex;               // queues, events, and the one kernel or copy that runs at a time in this mode
std::vector<hipemuStream*> allStreams;
hipemuStream nullStream;
uint64_t asyncState = asyncSeed * 0x9E3779B97F4A7C15ULL + 1;
std::vector<std::pair<const char*, size_t>> pinnedRanges;
std::mutex pinnedMutex;
bool isPinned(const void* p)
{
    std::lock_guard<std::mutex> lock(pinnedMutex);
    for(const auto& r : pinnedRanges) if(static_cast<const char*>(p) >= r.first && static_cast<const char*>(p) < r.first + r.second) return true;
    return false;
}
hipemuStream* streamOf(hipStream_t s) { return s ? s : &nullStream; }
void runFront(hipemuStream* s, int depth);
// Runs the stream's queue until `count` operations are left in it.
void drainTo(hipemuStream* s, size_t left, int depth = 0)
{
    if(depth > 64) { std::fprintf(stderr, "hipemu: HIPEMU_ASYNC: streams wait for each other's events in a circle\n"); std::abort(); }
    while(s->queue.size() > left) runFront(s, depth);
}
void runFront(hipemuStream* s, int depth)
{
    Op op = std::move(s->queue.front());
    s->queue.pop_front();
    if(op.waitsFor && op.waitsFor->completed < op.waitId) {
        // The record it waits for is still in another stream's queue: that stream runs up to it first.
        hipemuStream* other = op.waitsFor->stream;
        while(other && other != s && op.waitsFor->completed < op.waitId && !other->queue.empty()) runFront(other, depth + 1);
    }
    if(op.run) op.run();
    if(op.records) { op.records->completed = std::max(op.records->completed, op.recordId); op.records->t = std::chrono::steady_clock::now(); }
}
// Before a synchronisation delivers one stream's work, the others run up to a random point of their queues: the interleaving changes
// from call to call with the seed, and what is not asked for stays pending.
void disturbOthers(hipemuStream* except)
{
    for(hipemuStream* s : allStreams) {
        if(s == except || s->queue.empty()) continue;
        const uint64_t r = hipemu::nextGarbage(asyncState);
        if(r & 1u) drainTo(s, size_t((r >> 8) % (s->queue.size() + 1)));
    }
}
void enqueue(hipStream_t stream, Op op)
{
    std::lock_guard<std::recursive_mutex> lock(deviceMutex);
    hipemuStream* s = streamOf(stream);
    if(op.records) op.records->stream = s;
    s->queue.push_back(std::move(op));
}
void synchronizeStream(hipStream_t stream)
{
    if(!asyncSeed) return;
    std::lock_guard<std::recursive_mutex> lock(deviceMutex);
    disturbOthers(streamOf(stream));
    drainTo(streamOf(stream), 0);
}
void synchronizeDevice()
{
    if(!asyncSeed) return;
    std::lock_guard<std::recursive_mutex> lock(deviceMutex);
    // (in a random order of the streams; a stream that waits for another's event pulls that one along)
    std::vector<hipemuStream*> order(allStreams);
    order.push_back(&nullStream);
    for(size_t k = order.size(); k > 1; k--) std::swap(order[k - 1], order[hipemu::nextGarbage(asyncState) % k]);
    for(hipemuStream* s : order) drainTo(s, 0);
}
}  // namespace

namespace hipemu {
void launchOn(hipStream_t stream, const Launch& L, void* (*copy)(const void*), void (*destroy)(void*))
{
    if(!asyncSeed) { launch(L); return; }
    countLaunch();
    Launch own = L;
    own.args = copy(L.args);
    Op op;
    op.run = [own, destroy]() { try { launch(own); } catch(...) { destroy(own.args); throw; } destroy(own.args); };
    enqueue(stream, std::move(op));
}
}  // namespace hipemu

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int)
{
    std::memset(p, 0, sizeof(*p));
    std::snprintf(p->name, sizeof(p->name), "CPU emulation of wave64 (test infrastructure)");
    std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:emulated-on-cpu");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = size_t(16) << 30;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() { counted(N_DEVICE_SYNC); synchronizeDevice(); return hipSuccess; }
// Device memory comes back POISONED (0xA5 in every byte): a kernel that reads what no kernel, copy or memset wrote -- which fresh
// pages of the host, zero by the operating system's doing, would forgive, and a GPU's reused memory does not -- computes nonsense
// here too.  HIPEMU_NO_POISON=1 leaves the bytes as the allocator gives them.
hipError_t hipMalloc(void** p, size_t n)
{
    counted(N_MALLOC);
    const size_t bytes = (n + 255) / 256 * 256 + 256;
    *p = std::aligned_alloc(256, bytes);
    static const bool poison = [] { const char* e = std::getenv("HIPEMU_NO_POISON"); return !(e && e[0] == '1'); }();
    if(*p && poison) std::memset(*p, 0xA5, bytes);
    return *p ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipFree(void* p) { counted(N_FREE); synchronizeDevice(); std::free(p); return hipSuccess; }       // (hipFree waits for the device)
hipError_t hipHostMalloc(void** p, size_t n, unsigned)
{
    counted(N_HOST_MALLOC);
    *p = std::malloc(n ? n : 1);
    if(*p) { std::lock_guard<std::mutex> lock(pinnedMutex); pinnedRanges.emplace_back(static_cast<const char*>(*p), n ? n : 1); }
    return *p ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipHostFree(void* p)
{
    counted(N_HOST_FREE);
    synchronizeDevice();
    { std::lock_guard<std::mutex> lock(pinnedMutex); for(size_t k = 0; k < pinnedRanges.size(); k++) if(pinnedRanges[k].first == p) { pinnedRanges.erase(pinnedRanges.begin() + long(k)); break; } }
    std::free(p);
    return hipSuccess;
}
namespace {
hipError_t copyOn(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t stream, bool blocking)
{
    countedCopy(k);
    if(k == hipMemcpyDeviceToHost && !isPinned(d)) counted(N_COPY_D2H_PAGEABLE);      // (these block the calling thread by themselves)
    if(!asyncSeed) { if(n) std::memmove(d, s, n); return hipSuccess; }
    const bool fromHost = k == hipMemcpyHostToDevice || (k == hipMemcpyDefault && false);
    const bool toHost = k == hipMemcpyDeviceToHost;
    Op op;
    // (HIPEMU_ASYNC_STAGE=0: a pageable source is NOT staged but read when the copy runs, as the runtime does when it pins a large
    // source in place instead: host code that reuses or frees the source before a synchronisation then shows -- under AddressSanitizer
    // as a use after free)
    static const bool stage = [] { const char* e = std::getenv("HIPEMU_ASYNC_STAGE"); return !(e && e[0] == '0'); }();
    if(fromHost && stage && !isPinned(s)) {
        // pageable source: staged now, delivered when the stream gets there
        std::shared_ptr<std::vector<char>> staged = std::make_shared<std::vector<char>>(static_cast<const char*>(s), static_cast<const char*>(s) + n);
        op.run = [d, staged]() { if(!staged->empty()) std::memcpy(d, staged->data(), staged->size()); };
    } else {
        op.run = [d, s, n]() { if(n) std::memmove(d, s, n); };
    }
    enqueue(stream, std::move(op));
    if(blocking || (toHost && !isPinned(d))) synchronizeStream(stream);      // (a copy to pageable memory is complete when the call returns)
    return hipSuccess;
}
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k) { counted(N_STREAM_SYNC); return copyOn(d, s, n, k, nullptr, true); }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t stream) { return copyOn(d, s, n, k, stream, false); }
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t stream) { return copyOn(d, s, n, hipMemcpyDeviceToDevice, stream, false); }
hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
namespace {
hipError_t memsetOn(void* d, int v, size_t n, hipStream_t stream, bool blocking)
{
    counted(N_MEMSET);
    if(!asyncSeed) { if(n) std::memset(d, v, n); return hipSuccess; }
    Op op;
    op.run = [d, v, n]() { if(n) std::memset(d, v, n); };
    enqueue(stream, std::move(op));
    if(blocking) synchronizeStream(stream);
    return hipSuccess;
}
}
hipError_t hipMemset(void* d, int v, size_t n) { return memsetOn(d, v, n, nullptr, true); }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t stream) { return memsetOn(d, v, n, stream, false); }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned)
{
    counted(N_STREAM_CREATE);
    *s = new hipemuStream;
    std::lock_guard<std::recursive_mutex> lock(deviceMutex);
    allStreams.push_back(*s);
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
    synchronizeStream(s);
    { std::lock_guard<std::recursive_mutex> lock(deviceMutex); allStreams.erase(std::remove(allStreams.begin(), allStreams.end(), s), allStreams.end()); }
    delete s;
    return hipSuccess;
}
// HIPEMU_TRACE_SYNC=1: where the host synchronises from -- "hipemu: sync <offset in the library>" on stderr, for addr2line (scripts/emu_api_counts.py
// counts them; this says which they are).
hipError_t hipStreamSynchronize(hipStream_t s)
{
    counted(N_STREAM_SYNC);
    static const bool trace = [] { const char* e = std::getenv("HIPEMU_TRACE_SYNC"); return e && e[0] == '1'; }();
    if(trace) {
        void* const from = __builtin_return_address(0);
        Dl_info info;
        if(dladdr(from, &info) && info.dli_fbase) std::fprintf(stderr, "hipemu: sync %#zx\n", size_t(static_cast<char*>(from) - static_cast<char*>(info.dli_fbase)));
    }
    synchronizeStream(s);
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    counted(N_STREAM_WAIT);
    if(!asyncSeed) return hipSuccess;
    std::lock_guard<std::recursive_mutex> lock(deviceMutex);
    Op op;
    op.waitsFor = e; op.waitId = e->recorded;          // (the record made last BEFORE this call: a later one does not count)
    enqueue(s, std::move(op));
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) { counted(N_EVENT_CREATE); *e = new hipemuEvent; (*e)->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e)
{
    if(asyncSeed) { std::lock_guard<std::recursive_mutex> lock(deviceMutex); if(e->stream && e->completed < e->recorded) drainTo(e->stream, 0); }
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    counted(N_EVENT_RECORD);
    if(!asyncSeed) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
    std::lock_guard<std::recursive_mutex> lock(deviceMutex);
    Op op;
    op.records = e; op.recordId = ++e->recorded;
    enqueue(s, std::move(op));
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
    counted(N_EVENT_SYNC);
    if(!asyncSeed) return hipSuccess;
    std::lock_guard<std::recursive_mutex> lock(deviceMutex);
    const uint64_t want = e->recorded;
    if(e->completed < want && e->stream) { disturbOthers(e->stream); while(e->completed < want && !e->stream->queue.empty()) runFront(e->stream, 0); }
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e)
{
    if(!asyncSeed) return hipSuccess;
    std::lock_guard<std::recursive_mutex> lock(deviceMutex);
    return e->completed >= e->recorded ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
    if(asyncSeed) {
        std::lock_guard<std::recursive_mutex> lock(deviceMutex);
        if(a->completed < a->recorded || b->completed < b->recorded) return hipErrorNotReady;
    }
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
