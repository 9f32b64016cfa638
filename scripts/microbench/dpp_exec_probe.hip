// What a DPP move returns when its SOURCE lane is switched off by EXEC (gfx950), for the forward DP's plan to run lanes
// without a diagonal switched off instead of masking their values.  Lanes with (lane % 4 == 3) leave
// before the DPP instruction; every other lane reads its neighbour below and above, with bound_ctrl on (zero fill) and off
// (keep `old` = 0x55).  Prints one JSON line per variant with the 64 values.
//     hipcc --offload-arch=gfx950 -O2 -o shasta_amd/_build/dpp_exec_probe scripts/microbench/dpp_exec_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void probe(int32_t* out)
{
    const int lane = threadIdx.x;
    const int32_t v = 1000 + lane;
    int32_t r[4] = {-1, -1, -1, -1};
    uint64_t votes = 0;
    if(lane % 4 != 3) {
        r[0] = __builtin_amdgcn_update_dpp(0x55, v, 0x111, 0xf, 0xf, true);      // row_shr:1, bound_ctrl
        r[1] = __builtin_amdgcn_update_dpp(0x55, v, 0x111, 0xf, 0xf, false);     // row_shr:1
        r[2] = __builtin_amdgcn_update_dpp(0x55, v, 0x101, 0xf, 0xf, true);      // row_shl:1, bound_ctrl
        r[3] = __builtin_amdgcn_update_dpp(0x55, v, 0x130, 0xf, 0xf, true);      // wave_shl:1, bound_ctrl
        votes = __builtin_amdgcn_ballot_w64(true);
    }
    for(int k = 0; k < 4; k++) out[k * 64 + lane] = r[k];
    if(lane == 0) { out[256] = int32_t(votes); out[257] = int32_t(votes >> 32); }
}

int main()
{
    int32_t* d = nullptr;
    if(hipMalloc(&d, 258 * sizeof(int32_t)) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, nullptr, d);
    int32_t h[258];
    if(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const char* names[4] = {"row_shr:1 bound_ctrl", "row_shr:1", "row_shl:1 bound_ctrl", "wave_shl:1 bound_ctrl"};
    for(int k = 0; k < 4; k++) {
        std::printf("{\"variant\": \"%s\", \"old\": 85, \"source_value\": \"1000 + lane\", \"lanes_off\": \"lane %% 4 == 3\", \"values\": [", names[k]);
        for(int l = 0; l < 64; l++) std::printf("%d%s", h[k * 64 + l], l == 63 ? "" : ", ");
        std::printf("]}\n");
    }
    std::printf("{\"ballot_of_true_in_the_branch\": \"0x%08x%08x\"}\n", uint32_t(h[257]), uint32_t(h[256]));
    return 0;
}
