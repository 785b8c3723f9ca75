// LayerNormalization forward/backward for the ViT residual stream (gfx950).
// HBM-bound: one wave owns one token row (D <= 1024 fp32 values = <= 4 float4 per lane),
// statistics by wave shuffles, no LDS in the forward.  The residual stream is fp32, the
// normalised output that feeds the MFMA GEMMs is bf16.
//
// Replaces tf.keras.layers.LayerNormalization(epsilon=1e-6) constructed at
// chambers/layers/transformer.py:39,49,283 (moments + batch_normalization path).
#include "common.hpp"
#include "../../include/chambers_hip.h"

namespace {

// Streaming loads (CHB_LN_STREAM, NT): x, dy and the old dx are read once here and not again before they have left the caches;
// a non-temporal load does not allocate them on the way in, so it does not push out what the producer GEMM left there
// (tools/ln_bench.py, profiles/r04_layernorm_stream_ab.txt).  A/B builds: -DCHB_LN_BLOCKS=<n> caps the backward grid,
// -DCHB_LN_NO_TAIL drops the dgamma / dbeta / column-sum atomics (timing only).
template <bool NT>
__device__ __forceinline__ float4 ln_load(const float4* p) {
    if (!NT) return *p;
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <bool NT>
__device__ __forceinline__ uint2 ln_load(const uint2* p) {
    if (!NT) return *p;
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const u2 v = __builtin_nontemporal_load(reinterpret_cast<const u2*>(p));
    return make_uint2(v.x, v.y);
}
#ifndef CHB_LN_BLOCKS
#define CHB_LN_BLOCKS 1024
#endif

template <int NCH, bool NT>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const float* __restrict__ x, int64_t x_stride, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int M, int D,
                                                     float eps) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nchunk = D >> 2;
    const float inv_d = 1.0f / (float)D;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * x_stride);
        float4 v[NCH];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane + 64 * j;
            v[j] = (c < nchunk) ? ln_load<NT>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane + 64 * j;
            if (c < nchunk) {
                const float a = v[j].x - mean, b = v[j].y - mean, cc = v[j].z - mean, d = v[j].w - mean;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float var = wave_sum(q) * inv_d;
        const float rstd = 1.0f / sqrtf(var + eps);
        if (lane == 0) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
        uint2* yr = reinterpret_cast<uint2*>(y + (int64_t)row * D);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane + 64 * j;
            if (c < nchunk) {
                const float4 g = reinterpret_cast<const float4*>(gamma)[c];
                const float4 b = reinterpret_cast<const float4*>(beta)[c];
                uint2 o;
                o.x = pack_bf16x2((v[j].x - mean) * rstd * g.x + b.x, (v[j].y - mean) * rstd * g.y + b.y);
                o.y = pack_bf16x2((v[j].z - mean) * rstd * g.z + b.z, (v[j].w - mean) * rstd * g.w + b.w);
                yr[c] = o;
            }
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; dgamma += dy*xhat; dbeta += dy.
template <int NCH, bool NT>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x, int64_t x_stride,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, float* __restrict__ dx, int64_t dx_stride,
                                                     int accumulate, float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                     int D, bf16_t* __restrict__ dz, float* __restrict__ dzsum, float drop_scale,
                                                     uint32_t drop_thr, uint32_t drop_key, int zero_gaps) {
    __shared__ float red[3][4][NCH * 256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nchunk = D >> 2;
    const float inv_d = 1.0f / (float)D;
    float4 gam[NCH], dg[NCH], db[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = lane + 64 * j;
        gam[j] = (c < nchunk) ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 dzs[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) dzs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * x_stride);
        const uint2* dyr = reinterpret_cast<const uint2*>(dy + (int64_t)row * D);
        const float mean = mean_in[row], rstd = rstd_in[row];
        float4 xh[NCH], gy[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane + 64 * j;
            if (c < nchunk) {
                const float4 xv = ln_load<NT>(xr + c);
                const uint2 dv = ln_load<NT>(dyr + c);
                const float d0 = bf16_to_f32((bf16_t)(dv.x & 0xffff)), d1 = bf16_to_f32((bf16_t)(dv.x >> 16));
                const float d2 = bf16_to_f32((bf16_t)(dv.y & 0xffff)), d3 = bf16_to_f32((bf16_t)(dv.y >> 16));
                xh[j] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
                gy[j] = make_float4(d0 * gam[j].x, d1 * gam[j].y, d2 * gam[j].z, d3 * gam[j].w);
                s1 += (gy[j].x + gy[j].y) + (gy[j].z + gy[j].w);
                s2 += (gy[j].x * xh[j].x + gy[j].y * xh[j].y) + (gy[j].z * xh[j].z + gy[j].w * xh[j].w);
                dg[j].x += d0 * xh[j].x; dg[j].y += d1 * xh[j].y; dg[j].z += d2 * xh[j].z; dg[j].w += d3 * xh[j].w;
                db[j].x += d0; db[j].y += d1; db[j].z += d2; db[j].w += d3;
            } else {
                xh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                gy[j] = xh[j];
            }
        }
        const float c1 = wave_sum(s1) * inv_d, c2 = wave_sum(s2) * inv_d;
        float4* dxr = reinterpret_cast<float4*>(dx + (int64_t)row * dx_stride);
        // dz is a compact [rows, D] matrix over the SAME element grid as dx: with strided rows (class rows of the token matrix,
        // dx_stride = N * D) LayerNorm row `row` is token-matrix row row * N
        const int64_t zrow = (int64_t)row * (dx_stride / D);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane + 64 * j;
            if (c < nchunk) {
                float4 o = make_float4(rstd * (gy[j].x - c1 - xh[j].x * c2), rstd * (gy[j].y - c1 - xh[j].y * c2),
                                       rstd * (gy[j].z - c1 - xh[j].z * c2), rstd * (gy[j].w - c1 - xh[j].w * c2));
                if (accumulate) {
                    const float4 p = ln_load<NT>(dxr + c);
                    o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
                }
                dxr[c] = o;
                if (dz) {
                    // backward of the keras Dropout that follows in the backward chain: dz = dx * keep / (1-rate), bf16 GEMM
                    // operand; its column sums are the bias gradient of that GEMM's layer
                    float4 z = o;
                    if (drop_thr) {
                        const uint64_t e = (uint64_t)zrow * (uint64_t)D + (uint64_t)(c * 4);
                        bool k0, k1, k2, k3;
                        chb_keep2((uint32_t)(e >> 1), drop_key, drop_thr, k0, k1);
                        chb_keep2((uint32_t)(e >> 1) + 1u, drop_key, drop_thr, k2, k3);
                        z.x = k0 ? z.x * drop_scale : 0.f; z.y = k1 ? z.y * drop_scale : 0.f;
                        z.z = k2 ? z.z * drop_scale : 0.f; z.w = k3 ? z.w * drop_scale : 0.f;
                    }
                    uint2 zb;
                    zb.x = pack_bf16x2(z.x, z.y);
                    zb.y = pack_bf16x2(z.z, z.w);
                    reinterpret_cast<uint2*>(dz + zrow * D)[c] = zb;
                    // sum what the GEMM will read (the bf16-rounded values), as the separate colsum pass did
                    dzs[j].x += bf16_to_f32((bf16_t)(zb.x & 0xffff)); dzs[j].y += bf16_to_f32((bf16_t)(zb.x >> 16));
                    dzs[j].z += bf16_to_f32((bf16_t)(zb.y & 0xffff)); dzs[j].w += bf16_to_f32((bf16_t)(zb.y >> 16));
                }
            }
        }
    }
    if (zero_gaps) {
        // strided rows (the class rows of the token matrix: dx_stride = N * D): this launch also owns the floats between them.
        // Row r's gap is dx[r * dx_stride + D .. (r + 1) * dx_stride); it is cut into 64 KiB pieces dealt over the whole grid.
        const int64_t gap4 = (dx_stride - D) >> 2;
        const int64_t parts = (gap4 + 4095) >> 12;
        const int64_t items = (int64_t)M * parts;
        for (int64_t it = blockIdx.x; it < items; it += gridDim.x) {
            const int64_t row = it / parts, part = it - row * parts;
            float4* base = reinterpret_cast<float4*>(dx + row * dx_stride + D) + (part << 12);
            const int64_t cnt = min((int64_t)4096, gap4 - (part << 12));
            for (int64_t k = threadIdx.x; k < cnt; k += 256) base[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dz) {     // the same gap of the bf16 dz matrix: dropout-backward of a zero gradient (half the bytes: 8 elements per float4)
                float4* zb = reinterpret_cast<float4*>(dz + row * dx_stride + D) + (part << 11);
                const int64_t zc = min((int64_t)2048, (gap4 >> 1) - (part << 11));
                for (int64_t k = threadIdx.x; k < zc; k += 256) zb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
#ifdef CHB_LN_NO_TAIL
    if (M >= 0) return;
#endif
    // block-level reduce of dgamma/dbeta partials over the 4 waves, then one atomic per column
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int base = (lane + 64 * j) * 4;
        red[0][wave][base + 0] = dg[j].x; red[0][wave][base + 1] = dg[j].y;
        red[0][wave][base + 2] = dg[j].z; red[0][wave][base + 3] = dg[j].w;
        red[1][wave][base + 0] = db[j].x; red[1][wave][base + 1] = db[j].y;
        red[1][wave][base + 2] = db[j].z; red[1][wave][base + 3] = db[j].w;
        red[2][wave][base + 0] = dzs[j].x; red[2][wave][base + 1] = dzs[j].y;
        red[2][wave][base + 2] = dzs[j].z; red[2][wave][base + 3] = dzs[j].w;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        const float g = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        const float b = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
        atomicAdd(dgamma + c, g);
        atomicAdd(dbeta + c, b);
        if (dzsum) atomicAdd(dzsum + c, (red[2][0][c] + red[2][1][c]) + (red[2][2][c] + red[2][3][c]));
    }
}

inline int ln_grid(int M) {
    int blocks = (M + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    return blocks < 1 ? 1 : blocks;
}

}  // namespace

extern "C" {

int chb_layernorm_fwd(const float* x, int64_t x_stride, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      int M, int D, float eps, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || M < 0 || D <= 0) return CHB_EINVAL;
    if ((D & 3) || D > 1024 || (x_stride & 3)) return CHB_EUNSUPPORTED;
    if (M == 0) return CHB_OK;
    const int nch = (D / 4 + 63) / 64;
    const dim3 grid(ln_grid(M)), block(256);
    hipStream_t s = (hipStream_t)stream;
    bf16_t* yo = (bf16_t*)y;
    const bool nt = chb_option(CHB_OPT_LN_STREAM) != 0;
#define LN_FWD(NCH)                                                                                                              \
    do {                                                                                                                         \
        if (nt) hipLaunchKernelGGL((ln_fwd_kernel<NCH, true>), grid, block, 0, s, x, x_stride, gamma, beta, yo, mean, rstd, M, D, eps);  \
        else hipLaunchKernelGGL((ln_fwd_kernel<NCH, false>), grid, block, 0, s, x, x_stride, gamma, beta, yo, mean, rstd, M, D, eps);    \
    } while (0)
    switch (nch) {
        case 1: LN_FWD(1); break;
        case 2: LN_FWD(2); break;
        case 3: LN_FWD(3); break;
        default: LN_FWD(4); break;
    }
#undef LN_FWD
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_layernorm_bwd(const void* dy, const float* x, int64_t x_stride, const float* mean, const float* rstd, const float* gamma,
                      float* dx, int64_t dx_stride, int accumulate, float* dgamma, float* dbeta, int M, int D, void* dz_bf16,
                      float* dz_colsum, float drop_rate, uint32_t drop_key, int zero_gaps, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || M < 0 || D <= 0) return CHB_EINVAL;
    if (drop_rate < 0.f || drop_rate >= 1.f) return CHB_EINVAL;
    if (zero_gaps && (accumulate || dx_stride < D)) return CHB_EINVAL;
    // the fused dropout-backward tail writes a compact dz over dx's element grid: strided rows need whole-row strides, and the rows
    // between them are only defined (as zeros) when this launch also fills the gaps
    if (dz_bf16 && dx_stride != D && (!zero_gaps || dx_stride % D != 0 || ((dx_stride - D) & 7) || (D & 7))) return CHB_EINVAL;
    bf16_t* dz = (bf16_t*)dz_bf16;
    const float dscale = 1.0f / (1.0f - drop_rate);
    const uint32_t dthr = drop_rate > 0.f ? chb_drop_threshold(drop_rate) : 0u;
    if ((D & 3) || D > 1024 || (x_stride & 3) || (dx_stride & 3)) return CHB_EUNSUPPORTED;
    if (M == 0) return CHB_OK;
    const int nch = (D / 4 + 63) / 64;
    int blocks = (M + 3) / 4;
    if (blocks > CHB_LN_BLOCKS || (zero_gaps && dx_stride > D)) blocks = CHB_LN_BLOCKS;    // the gap fill wants the whole chip
    const dim3 grid(blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bf16_t* d = (const bf16_t*)dy;
    const bool nt = chb_option(CHB_OPT_LN_STREAM) != 0;
#define LN_BWD(NCH)                                                                                                              \
    do {                                                                                                                         \
        if (nt) hipLaunchKernelGGL((ln_bwd_kernel<NCH, true>), grid, block, 0, s, d, x, x_stride, mean, rstd, gamma, dx, dx_stride, accumulate, dgamma, dbeta, M, D, dz, dz_colsum, dscale, dthr, drop_key, zero_gaps);  \
        else hipLaunchKernelGGL((ln_bwd_kernel<NCH, false>), grid, block, 0, s, d, x, x_stride, mean, rstd, gamma, dx, dx_stride, accumulate, dgamma, dbeta, M, D, dz, dz_colsum, dscale, dthr, drop_key, zero_gaps); \
    } while (0)
    switch (nch) {
        case 1: LN_BWD(1); break;
        case 2: LN_BWD(2); break;
        case 3: LN_BWD(3); break;
        default: LN_BWD(4); break;
    }
#undef LN_BWD
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

}  // extern "C"
