// Stand-in for a collective's kernel in single-GPU contention experiments (tools/rccl_contention.py): n_wg workgroups of 256
// threads, 16 KiB of LDS each, that hold their CUs for `micros` microseconds and touch no memory.  Not part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) hog_kernel(int ticks, int* sink) {
    __shared__ int pad[4096];
    pad[threadIdx.x] = threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    while ((int64_t)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks < 0) sink[0] = pad[(threadIdx.x + 1) & 255];         // never taken: keeps the LDS allocation
}

extern "C" int hog_launch(int n_wg, int micros, void* stream) {
    if (n_wg <= 0 || micros <= 0) return -1;
    hipLaunchKernelGGL(hog_kernel, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, micros * 100, (int*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
