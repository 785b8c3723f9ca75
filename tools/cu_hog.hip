// Stand-in for a collective's kernel in single-GPU contention experiments (tools/rccl_contention.py): n_wg workgroups of 256
// threads, 16 KiB of LDS each (so they cannot share a CU with a persistent GEMM workgroup, which holds all 160 KiB), that hold
// their CUs for `micros` microseconds and meanwhile stream a buffer (read + write, like the reduce-scatter / all-gather passes of a
// ring all-reduce over its bucket).  Not part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) hog_kernel(int ticks, float4* buf, long n_vec) {
    __shared__ int pad[4096];
    pad[threadIdx.x] = threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    const long per = n_vec / gridDim.x;
    float4* mine = buf + (long)blockIdx.x * per;
    long i = threadIdx.x;
    while ((int64_t)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) {
        if (buf && per > 0) {
            float4 v = mine[i % per];
            v.x += 1.0f;
            mine[i % per] = v;
            i += 256;
        } else {
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if (ticks < 0) buf[0].x = (float)pad[(threadIdx.x + 1) & 255];         // never taken: keeps the LDS allocation
}

extern "C" int hog_launch(int n_wg, int micros, void* buf, long bytes, void* stream) {
    if (n_wg <= 0 || micros <= 0) return -1;
    hipLaunchKernelGGL(hog_kernel, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, micros * 100, (float4*)buf, bytes / 16);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
