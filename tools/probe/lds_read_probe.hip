// Hardware probe (not part of the library): which lane -> address patterns does ds_read_b128 serve without bank conflicts on gfx950?
// Every wave issues the 16-row x 4-chunk fragment read of a 16x16x32 bf16 MFMA operand (lane = (g, i): row i, 16-byte k-chunk g)
// against a few tile layouts; time per read instruction, 4 waves per CU.  hipcc --offload-arch=gfx950 -O3 -o lds_read_probe lds_read_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) uint32_t u4;

__device__ __forceinline__ uint32_t lane_addr(int v, int g, int i) {
    switch (v) {
        case 0: return (uint32_t)((g * 16 + i) * 16);                                   // linear: lane * 16
        case 1: return (uint32_t)(i * 128 + ((g ^ ((i >> 1) & 7)) << 4));               // 128-byte rows, chunk ^ ((r >> 1) & 7)  (gemm_nt256sp)
        case 2: return (uint32_t)(i * 64 + ((g ^ ((i >> 2) & 3)) << 4));                // 64-byte rows, chunk ^ ((r >> 2) & 3)
        case 3: return (uint32_t)(i * 64 + ((g ^ (i & 3)) << 4));                       // 64-byte rows, chunk ^ (r & 3)
        case 4: return (uint32_t)(i * 64 + ((g ^ ((i >> 1) & 3)) << 4));                // 64-byte rows, chunk ^ ((r >> 1) & 3)
        case 5: return (uint32_t)(i * 64 + (g << 4));                                   // 64-byte rows, no swizzle
        case 6: return (uint32_t)((i >> 1) * 128 + ((((i & 1) * 4 + g) ^ ((i >> 2) & 7)) << 4));   // row pairs as 128-byte rows, slot ^ ((pair >> 1) & 7)
        case 7: return (uint32_t)((i >> 1) * 128 + ((((i & 1) * 4 + g) ^ ((i >> 1) & 7)) << 4));   // row pairs, slot ^ (pair & 7)
        case 8: return (uint32_t)(i * 64 + (((g + (i >> 2)) & 3) << 4));                // 64-byte rows, chunk + (r >> 2)
        case 9: return (uint32_t)(i * 64 + ((g ^ ((i >> 3) & 1) ^ (((i >> 2) & 1) << 1)) << 4));   // 64-byte rows, bit-reversed (r >> 2)
        default: return (uint32_t)(i * 128 + (g << 4));                                 // 128-byte rows, no swizzle
    }
}

__global__ void __launch_bounds__(256) ldsread(int v, int iters, unsigned long long* out, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    for (int k = threadIdx.x; k < 16384; k += 256) reinterpret_cast<uint32_t*>(lds)[k] = k;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)lds) + wave * 16384 + lane_addr(v, g, i);
    u4 r0, r1, r2, r3, r4, r5, r6, r7;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("ds_read_b128 %0, %8 offset:0\n\tds_read_b128 %1, %8 offset:2048\n\tds_read_b128 %2, %8 offset:4096\n\tds_read_b128 %3, %8 offset:6144\n\t"
                     "ds_read_b128 %4, %8 offset:8192\n\tds_read_b128 %5, %8 offset:10240\n\tds_read_b128 %6, %8 offset:12288\n\tds_read_b128 %7, %8 offset:14336\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(base));
        acc += r0.x + r1.y + r2.z + r3.w + r4.x + r5.y + r6.z + r7.w;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

int main() {
    unsigned long long* out;
    uint32_t* sink;
    CHECK(hipMalloc(&out, 256 * 4 * 8));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipFuncSetAttribute((const void*)ldsread, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const char* names[] = {"linear (lane * 16)", "128-byte rows, chunk ^ ((r >> 1) & 7)   [gemm_nt256sp]", "64-byte rows, chunk ^ ((r >> 2) & 3)", "64-byte rows, chunk ^ (r & 3)",
                           "64-byte rows, chunk ^ ((r >> 1) & 3)", "64-byte rows, no swizzle", "row pairs as 128-byte rows, slot ^ ((pair >> 1) & 7)",
                           "row pairs as 128-byte rows, slot ^ (pair & 7)", "64-byte rows, chunk + (r >> 2)", "64-byte rows, chunk ^ bitrev(r >> 2)", "128-byte rows, no swizzle"};
    const int iters = 4096;
    for (int v = 0; v < 11; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(ldsread, dim3(256), dim3(256), 65536, 0, v, iters, out, sink);
            CHECK(hipDeviceSynchronize());
        }
        static unsigned long long h[1024];
        CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        double s = 0;
        for (int k = 0; k < 1024; ++k) s += (double)h[k];
        printf("%-62s %6.1f clocks per ds_read_b128 and wave (4 waves per CU reading)\n", names[v], s / 1024.0 / (iters * 8.0));
    }
    return 0;
}
