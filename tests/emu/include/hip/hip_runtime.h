// TEST INFRASTRUCTURE ONLY -- never part of the product, never a fallback.
//
// A stand-in for <hip/hip_runtime.h> that lets the UNMODIFIED kernel sources of
// shasta_amd/csrc/*.hip be compiled by g++ and executed on CPU cores, wave64 semantics intact:
// every work-item of a workgroup is a fiber; the 64 fibers of a wavefront meet at every
// cross-lane operation (__ballot, __shfl*, __any, readlane, wave_barrier) and the workgroup's
// fibers meet at __syncthreads.  Between two such points lanes run one after the other, which
// is a legal schedule of the hardware's lock-step execution for code that orders its LDS
// traffic with those operations (the kernels do: waveLdsSync / __syncthreads).
// Purpose: run the parity tests of the kernel SOURCE on a machine without a GPU
// (tests/test_emu_*.py; SHASTA_EMU=1 pytest -m gpu).  It says nothing about performance, LDS
// capacity, or memory-model races between workgroups -- that is what the MI355X run is for.
// The product library (shasta_amd/_build) is built by hipcc and never sees this header.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <type_traits>
#include <utility>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ thread_local            // block scope: one copy per OS thread = per resident workgroup
#define HIP_SYMBOL(x) x
#ifdef __clang__
// clang knows __hip_atomic_load as a builtin in every language mode; its scope argument is checked.
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
#else
#define __HIP_MEMORY_SCOPE_AGENT 0
#endif

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x = 1, unsigned y = 1, unsigned z = 1) : x(x), y(y), z(z) {}
};
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

// ---------------------------------------------------------------------------
// Runtime API (host side).  One "device"; streams and events are ordering-free because every
// launch and copy completes before it returns.
// ---------------------------------------------------------------------------
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
constexpr hipError_t hipErrorNotReady = 600;
constexpr hipError_t hipErrorPeerAccessAlreadyEnabled = 704;
constexpr hipError_t hipErrorLaunchFailure = 719;
typedef struct hipemuStream* hipStream_t;
typedef struct hipemuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
constexpr unsigned hipStreamNonBlocking = 1;
constexpr unsigned hipHostMallocDefault = 0;
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };

const char* hipGetErrorString(hipError_t);
hipError_t hipGetLastError();
hipError_t hipGetDeviceCount(int*);
hipError_t hipGetDevice(int*);
hipError_t hipSetDevice(int);
hipError_t hipGetDeviceProperties(hipDeviceProp_t*, int);
hipError_t hipDeviceSynchronize();
hipError_t hipMalloc(void**, size_t);
hipError_t hipFree(void*);
hipError_t hipHostMalloc(void**, size_t, unsigned);
hipError_t hipHostFree(void*);
hipError_t hipMemcpy(void*, const void*, size_t, hipMemcpyKind);
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipMemcpyPeerAsync(void*, int, const void*, int, size_t, hipStream_t);
hipError_t hipDeviceCanAccessPeer(int*, int, int);
hipError_t hipDeviceEnablePeerAccess(int, unsigned);
hipError_t hipMemset(void*, int, size_t);
hipError_t hipMemsetAsync(void*, int, size_t, hipStream_t);
hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned);
hipError_t hipEventCreate(hipEvent_t*);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventQuery(hipEvent_t);
hipError_t hipEventElapsedTime(float*, hipEvent_t, hipEvent_t);
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int);
template<class T> hipError_t hipMemcpyFromSymbol(void* dst, const T& symbol, size_t n) { std::memcpy(dst, &symbol, n); return hipSuccess; }
template<class T> hipError_t hipMemcpyToSymbol(T& symbol, const void* src, size_t n) { std::memcpy(&symbol, src, n); return hipSuccess; }

// ---------------------------------------------------------------------------
// Execution model.
// ---------------------------------------------------------------------------
namespace hipemu {

struct Index { unsigned x, y, z; };
struct Fiber {                     // one work-item
    void* sp;                      // saved stack pointer while switched out
    Index tIdx, bIdx, bDim, gDim;
    int lane, wave, state;
    int kind;                      // pending collective
    const void* site;              // its call site
    uint64_t value, aux, result;
};
extern thread_local Fiber* cur;

enum Kind { BALLOT = 1, SHUFFLE, FIRSTLANE, WAVE_BARRIER, BLOCK_BARRIER, DPP_MOVE };

// Blocks the calling fiber until its wavefront (workgroup for BLOCK_BARRIER) has arrived.
uint64_t collective(int kind, uint64_t value, uint64_t aux) __attribute__((noinline));

struct Launch {
    dim3 grid, block;
    size_t dynamicLdsBytes;
    void (*invoke)(void* args);
    void* args;
    const char* name = "?";            // the kernel as the launch site spells it, and its address (HIPEMU_PROFILE's key)
    const void* kernel = nullptr;
};
void launch(const Launch&);
// HIPEMU_ASYNC (hip_emu.cpp): the launch waits in its stream's queue; `copy` makes the arguments outlive the caller's frame.
void launchOn(hipStream_t stream, const Launch&, void* (*copy)(const void*), void (*destroy)(void*));

template<class T> __forceinline__ uint64_t toBits(T v)
{
    static_assert(sizeof(T) <= 8, "shuffle of a type wider than 8 bytes");
    uint64_t b = 0; std::memcpy(&b, &v, sizeof(T)); return b;
}
template<class T> __forceinline__ T fromBits(uint64_t b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; }

}  // namespace hipemu

#define threadIdx (hipemu::cur->tIdx)
#define blockIdx (hipemu::cur->bIdx)
#define blockDim (hipemu::cur->bDim)
#define gridDim (hipemu::cur->gDim)

template<class... P, class... A>
void hipLaunchKernelNamed(const char* name, void (*kernel)(P...), dim3 grid, dim3 block, size_t dynamicLdsBytes, hipStream_t stream, A&&... a)
{
    struct Call { void (*kernel)(P...); std::tuple<std::decay_t<P>...> args; } call{kernel, std::tuple<std::decay_t<P>...>{static_cast<std::decay_t<P>>(a)...}};
    hipemu::Launch l{grid, block, dynamicLdsBytes,
        [](void* p) { Call* c = static_cast<Call*>(p); std::apply(c->kernel, c->args); }, &call, name, reinterpret_cast<const void*>(kernel)};
    hipemu::launchOn(stream, l, [](const void* p) -> void* { return new Call(*static_cast<const Call*>(p)); }, [](void* p) { delete static_cast<Call*>(p); });
}
#define hipLaunchKernelGGL(kernel, ...) hipLaunchKernelNamed(#kernel, kernel, __VA_ARGS__)

// ---------------------------------------------------------------------------
// Device intrinsics.
// ---------------------------------------------------------------------------
__forceinline__ uint64_t __ballot(int predicate) { return hipemu::collective(hipemu::BALLOT, predicate ? 1 : 0, 0); }
__forceinline__ uint64_t __builtin_amdgcn_ballot_w64(bool predicate) { return hipemu::collective(hipemu::BALLOT, predicate ? 1 : 0, 0); }
__forceinline__ int __any(int predicate) { return __ballot(predicate) != 0; }
__forceinline__ int __all(int predicate) { return __ballot(!predicate) == 0; }
__forceinline__ void __syncthreads() { (void)hipemu::collective(hipemu::BLOCK_BARRIER, 0, 0); }
__forceinline__ void __builtin_amdgcn_wave_barrier() { (void)hipemu::collective(hipemu::WAVE_BARRIER, 0, 0); }
#define __builtin_amdgcn_fence(order, scope) do {} while(0)
__forceinline__ void __threadfence() {}

// aux = source lane (or 64 + own lane when the source is outside the segment: keep own value).
template<class T> __forceinline__ T __shfl(T v, int srcLane, int width = 64)
{
    const int lane = hipemu::cur->lane;
    const int src = (lane / width) * width + (((srcLane % width) + width) % width);
    return hipemu::fromBits<T>(hipemu::collective(hipemu::SHUFFLE, hipemu::toBits(v), uint64_t(src)));
}
template<class T> __forceinline__ T __shfl_up(T v, unsigned delta, int width = 64)
{
    const int lane = hipemu::cur->lane;
    const int src = (lane % width) >= int(delta) ? lane - int(delta) : lane;
    return hipemu::fromBits<T>(hipemu::collective(hipemu::SHUFFLE, hipemu::toBits(v), uint64_t(src)));
}
template<class T> __forceinline__ T __shfl_down(T v, unsigned delta, int width = 64)
{
    const int lane = hipemu::cur->lane;
    const int src = (lane % width) + int(delta) < width ? lane + int(delta) : lane;
    return hipemu::fromBits<T>(hipemu::collective(hipemu::SHUFFLE, hipemu::toBits(v), uint64_t(src)));
}
template<class T> __forceinline__ T __shfl_xor(T v, int mask, int width = 64)
{
    const int lane = hipemu::cur->lane;
    const int peer = lane ^ mask;
    const int src = (peer / width == lane / width) ? peer : lane;
    return hipemu::fromBits<T>(hipemu::collective(hipemu::SHUFFLE, hipemu::toBits(v), uint64_t(src)));
}
__forceinline__ uint32_t __builtin_amdgcn_readlane(uint32_t v, int lane)
{
    return uint32_t(hipemu::collective(hipemu::SHUFFLE, v, uint64_t(lane)));
}
// v_mov_b32_dpp with row_mask = bank_mask = 0xf: the controls the kernels use (shifts by one lane).
// A lane whose source is outside the row / wavefront keeps `old`, or gets 0 with bound_ctrl.
__forceinline__ int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int rowMask, int bankMask, bool boundCtrl)
{
    const int lane = hipemu::cur->lane;
    int from = -1;
    switch(ctrl) {
        case 0x111: from = (lane % 16) >= 1 ? lane - 1 : -1; break;      // row_shr:1
        case 0x101: from = (lane % 16) <= 14 ? lane + 1 : -1; break;     // row_shl:1
        case 0x138: from = lane >= 1 ? lane - 1 : -1; break;             // wave_shr:1
        case 0x130: from = lane <= 62 ? lane + 1 : -1; break;            // wave_shl:1
        case 0x142: from = lane >= 16 ? (lane / 16) * 16 - 1 : -1; break;    // row_bcast:15 (lane 15 of a row to every lane of the next row)
        case 0x143: from = lane >= 32 ? 31 : -1; break;                      // row_bcast:31 (lane 31 to rows 2 and 3)
        default:
            if(ctrl > 0x110 && ctrl <= 0x11f) { const int n = ctrl - 0x110; from = (lane % 16) >= n ? lane - n : -1; break; }      // row_shr:n
            std::fprintf(stderr, "hip_emu: DPP control 0x%x is not modelled\n", ctrl); std::abort();
    }
    if(bankMask != 0xf) { std::fprintf(stderr, "hip_emu: DPP bank masks are not modelled\n"); std::abort(); }
    // row_mask: a row (16 lanes) whose bit is clear keeps `old` (it still takes part in the exchange: other rows may read its lanes).
    const bool rowOff = ((rowMask >> (lane / 16)) & 1) == 0;
    if(rowOff) from = -1;
    // A source lane that is out of range OR switched off (not at this call site with the others) is invalid: 0 with bound_ctrl,
    // `old` without -- measured on gfx950 (scripts/microbench/dpp_exec_probe.hip, profiles/r02_dpp_exec_probe.jsonl).
    const uint64_t got = hipemu::collective(hipemu::DPP_MOVE, uint32_t(src), uint64_t(from < 0 ? lane : from));
    if(rowOff) return old;
    const bool valid = from >= 0 && (got >> 32) != 0;
    return valid ? int(uint32_t(got)) : (boundCtrl ? 0 : old);
}
__forceinline__ uint32_t __builtin_amdgcn_readfirstlane(uint32_t v)
{
    return uint32_t(hipemu::collective(hipemu::FIRSTLANE, v, 0));
}

__forceinline__ int __popc(unsigned v) { return __builtin_popcount(v); }
__forceinline__ int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
__forceinline__ int __ffs(int v) { return __builtin_ffs(v); }
__forceinline__ int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
__forceinline__ int __ffsll(long long v) { return __builtin_ffsll(v); }
__forceinline__ int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
__forceinline__ int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
__forceinline__ unsigned __brev(unsigned v)
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    return __builtin_bswap32(v);
}
__forceinline__ unsigned __umulhi(unsigned a, unsigned b) { return unsigned((uint64_t(a) * uint64_t(b)) >> 32); }
// v_alignbyte_b32: bytes [c, c+4) of the 8-byte value hi:lo.
__forceinline__ uint32_t __builtin_amdgcn_alignbyte(uint32_t hi, uint32_t lo, uint32_t c)
{
    return uint32_t(((uint64_t(hi) << 32) | uint64_t(lo)) >> (8u * (c & 3u)));
}
// s_store_dwordx4 + s_dcache_wb (inline assembly in primitives.hpp): lane 0 of the wavefront stores.
#define SHASTA_SCALAR_STORE_DEFINED 1
__forceinline__ void scalarStore128(void* address, uint64_t low, uint64_t high)
{
    if((uint32_t(threadIdx.x) & 63u) == 0) { uint64_t v[2] = {low, high}; std::memcpy(address, v, 16); }
}
__forceinline__ void scalarStore128At(void* address, int byteOffset, uint64_t low, uint64_t high) { scalarStore128(static_cast<char*>(address) + byteOffset, low, high); }
__forceinline__ void scalarStoreFlush() {}
template<class T> __forceinline__ T* uniformPointer(T* p) { return p; }
#ifndef __clang__
template<class T> __forceinline__ T __hip_atomic_load(const T* p, int, int) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
#endif

// Workgroups run on different OS threads: global atomics are real atomics (LDS ones need not be,
// but one implementation serves both).
namespace hipemu { template<class T> struct Same { using type = T; }; }
template<class T> __forceinline__ T atomicAdd(T* p, typename hipemu::Same<T>::type v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template<class T> __forceinline__ T atomicOr(T* p, typename hipemu::Same<T>::type v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template<class T> __forceinline__ T atomicMax(T* p, typename hipemu::Same<T>::type v)
{
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while(old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
template<class T> __forceinline__ T atomicMin(T* p, typename hipemu::Same<T>::type v)
{
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while(old > v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
template<class T> __forceinline__ T atomicCAS(T* p, typename hipemu::Same<T>::type expected, typename hipemu::Same<T>::type desired)
{
    __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return expected;
}

// HIP's device min / max accept mixed integer types.
template<class A, class B> __forceinline__ constexpr std::common_type_t<A, B> min(A a, B b)
{
    using T = std::common_type_t<A, B>;
    return T(b) < T(a) ? T(b) : T(a);
}
template<class A, class B> __forceinline__ constexpr std::common_type_t<A, B> max(A a, B b)
{
    using T = std::common_type_t<A, B>;
    return T(a) < T(b) ? T(b) : T(a);
}
