// Test / measurement helper (tests/test_kernels_gpu.py, tools/cu_contention.py), not part of the product: a kernel whose workgroups do nothing but hold
// their compute units for a bounded time, standing in for the RCCL all-reduce kernels that share the GPU with the
// backward sweep of the last micro-batch when data parallelism overlaps the gradient exchange.
#include <hip/hip_runtime.h>
__global__ void squat_kernel(unsigned long long ticks) {  // 100 MHz ticks (s_memrealtime)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
extern "C" int squat_launch(int blocks, int threads, long long ticks, void* stream) {
    if (blocks <= 0 || threads <= 0 || ticks <= 0 || ticks > 100000000LL) return -1;  // at most 1 s
    hipLaunchKernelGGL(squat_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (unsigned long long)ticks);
    return (int)hipGetLastError();
}
