// TEST INFRASTRUCTURE: what HIPEMU_ASYNC makes of streams and events (tests/test_emu_runtime.py builds and runs this against hip_emu.cpp).
// Prints "name value" lines; the test says what they must be with and without the switch.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void setKernel(int* p, int v) { if(threadIdx.x == 0) *p = v; }
__global__ void copyKernel(const int* a, int* b) { if(threadIdx.x == 0) *b = *a; }

int main()
{
    int *a, *b, *pinned;
    hipMalloc((void**)&a, 4); hipMalloc((void**)&b, 4); hipHostMalloc((void**)&pinned, 4, 0);
    *pinned = -1;
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipMemsetAsync(a, 0, 4, s1); hipMemsetAsync(b, 0, 4, s1); hipStreamSynchronize(s1);
    // A launch is pending until something synchronises its stream.
    hipLaunchKernelGGL(setKernel, dim3(1), dim3(64), 0, s1, a, 7);
    std::printf("value_before_synchronisation %d\n", *a);
    hipStreamSynchronize(s1);
    std::printf("value_after_synchronisation %d\n", *a);
    // Another stream that waits for an event sees what was recorded before it, and a copy to pinned memory is delivered by the synchronisation.
    hipEvent_t e; hipEventCreate(&e);
    hipLaunchKernelGGL(setKernel, dim3(1), dim3(64), 0, s1, a, 9);
    hipEventRecord(e, s1);
    hipStreamWaitEvent(s2, e, 0);
    hipLaunchKernelGGL(copyKernel, dim3(1), dim3(64), 0, s2, (const int*)a, b);
    hipMemcpyAsync(pinned, b, 4, hipMemcpyDeviceToHost, s2);
    std::printf("event_query_before %d\n", int(hipEventQuery(e) == hipSuccess));
    std::printf("pinned_before_synchronisation %d\n", *pinned);
    hipStreamSynchronize(s2);
    std::printf("pinned_after_synchronisation %d\n", *pinned);
    std::printf("event_query_after %d\n", int(hipEventQuery(e) == hipSuccess));
    // A copy to pageable memory is complete when the call returns; a pageable source is staged at the call.
    int host = -1, source = 5;
    hipMemcpyAsync(a, &source, 4, hipMemcpyHostToDevice, s1);
    source = 6;
    hipMemcpyAsync(&host, a, 4, hipMemcpyDeviceToHost, s1);
    std::printf("pageable_round_trip %d\n", host);
    // hipFree waits for the device.
    hipLaunchKernelGGL(setKernel, dim3(1), dim3(64), 0, s2, b, 11);
    int* c; hipMalloc((void**)&c, 4); hipFree(c);
    std::printf("value_after_free %d\n", *b);
    return 0;
}
