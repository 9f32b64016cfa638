// Issue rate of the integer instructions the window hash and the cells kernel lean on, measured on the device
// (gfx950): independent chains per lane, enough waves to fill every SIMD; prints wave-instructions per cycle per CU
// next to v_add_u32 (full rate = 4 SIMDs x 1 wave-instruction per 4 cycles... on the 16-lane SIMD of CDNA).
// Build: hipcc --offload-arch=gfx950 -O3 -o shasta_amd/_build/valu_rates scripts/microbench/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if(e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while(0)

constexpr int CHAINS = 8, ITER = 4096;

template<int OP>
__global__ void __launch_bounds__(256) rateKernel(uint32_t* out, uint32_t seed)
{
    uint32_t a[CHAINS];
    uint64_t w[CHAINS];
    for(int k = 0; k < CHAINS; k++) { a[k] = seed + threadIdx.x * 7u + k; w[k] = (uint64_t(a[k]) << 32) | (a[k] * 3u + 1u); }
    const uint32_t c = seed | 0x5bd1e995u;
    const uint32_t c2 = seed ^ 0xc6a4a793u;
    for(int i = 0; i < ITER; i++) {
#pragma unroll
        for(int k = 0; k < CHAINS; k++) {
            // Inline assembly: the optimiser would otherwise replace 4096 additions or multiplications by a closed form.
            if(OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
            if(OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
            if(OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
            if(OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[k]) : "v"(a[k]), "v"(c) : "vcc");
            if(OP == 4) {       // 64 x 64 -> 64 by a constant: mad_u64_u32 + two mul_lo + add3 (what the hash does per multiply)
                uint32_t lo = uint32_t(w[k]), hi = uint32_t(w[k] >> 32), t0, t1;
                uint64_t p;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(lo), "v"(c) : "vcc");
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(t0) : "v"(lo), "v"(c2));
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(t1) : "v"(hi), "v"(c));
                asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(hi) : "v"(uint32_t(p >> 32)), "v"(t0), "v"(t1));
                w[k] = (uint64_t(hi) << 32) | uint32_t(p);
            }
            if(OP == 5) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(c)); asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[k])); }
            if(OP == 6) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[k]) : "v"(c));
            if(OP == 7) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[k]) : "v"(c));
        }
    }
    uint32_t s = 0;
    for(int k = 0; k < CHAINS; k++) s ^= a[k] ^ uint32_t(w[k]) ^ uint32_t(w[k] >> 32);
    if(s == 0x12345u) out[0] = s;
}

template<int OP> int run(const char* name, double opsPerIteration, uint32_t* out, int cus, double ghz)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int blocks = cus * 8;
    hipLaunchKernelGGL(rateKernel<OP>, dim3(blocks), dim3(256), 0, nullptr, out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a, nullptr));
    hipLaunchKernelGGL(rateKernel<OP>, dim3(blocks), dim3(256), 0, nullptr, out, 12345u);
    CHECK(hipEventRecord(b, nullptr));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double waveOps = double(blocks) * 4 * CHAINS * ITER * opsPerIteration;
    const double cycles = ms * 1e-3 * ghz * 1e9;
    std::printf("{\"op\": \"%s\", \"ms\": %.4f, \"wave_instructions_per_cycle_per_cu\": %.4f, \"lane_ops_per_s\": %.4e}\n",
        name, ms, waveOps / cycles / cus, waveOps * 64 / (ms * 1e-3));
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f}\n", prop.gcnArchName, cus, ghz);
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&out, 4096));
    if(run<0>("v_add_u32", 1, out, cus, ghz)) return 1;
    if(run<1>("v_mul_lo_u32", 1, out, cus, ghz)) return 1;
    if(run<2>("v_mul_hi_u32", 1, out, cus, ghz)) return 1;
    if(run<3>("v_mad_u64_u32", 1, out, cus, ghz)) return 1;
    if(run<4>("mul64 = mad_u64_u32 + 2 mul_lo + add3", 1, out, cus, ghz)) return 1;
    if(run<5>("xor+shift pair", 2, out, cus, ghz)) return 1;
    if(run<6>("v_mul_u32_u24", 1, out, cus, ghz)) return 1;
    if(run<7>("v_mad_u32_u24", 1, out, cus, ghz)) return 1;
    return 0;
}
