// HBM-bound glue around the ViT block on gfx950: dropout masks, embedding-stage forward and
// backward pieces, bias-gradient column sums, sparse softmax cross-entropy, the per-step
// fp32 -> bf16 (+transposed) weight refresh and the fused AdamW update.
//
// Reference call sites: keras Dropout (vision_transformer.py:261, layers/transformer.py:38,48),
// ConcatEmbedding / LearnedEmbedding1D (layers/embedding.py:179-180,251-261),
// AdamW (optimizers.py:147-155,372-464).
#include "common.hpp"
#include "../../include/chambers_hip.h"
#include <atomic>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

namespace {

struct OptionDef { const char* name; int dflt; };
const OptionDef kOptions[CHB_OPT_COUNT] = {
    {"ATTN_FWD_ALGO", 0}, {"ATTN_BWD_ALGO", 0}, {"AFFINE_ALGO", 0}, {"GEMM_ALGO", 0}, {"GEMM_WALK", 1}, {"TN_ATOMICS", 0}, {"TN_FAST", 1},
    {"GEMM_TILE_QUEUE", 0}, {"LN_STREAM", 0}, {"DEBUG", 0},
};
std::atomic<int> g_option[CHB_OPT_COUNT];
std::atomic<bool> g_option_read[CHB_OPT_COUNT];

inline int grid_for(int64_t n, int cap = 8192) {
    int64_t b = (n + 255) / 256;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

__global__ void dropout_mask_kernel(uint8_t* out, int64_t n, uint32_t thr, uint32_t key) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        out[e] = chb_keep((uint64_t)e, key, thr) ? 1 : 0;
}

__global__ void cls_row_kernel(float* x, const float* cls, const float* pos, int B, int N, int D, int row, float scale, uint32_t thr,
                               uint32_t key) {
    const int64_t total = (int64_t)B * D;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(t / D), d = (int)(t - (int64_t)b * D);
        const int64_t e = ((int64_t)b * N + row) * D + d;
        float v = cls[d] + pos[(int64_t)row * D + d];
        if (thr) v = chb_keep((uint64_t)e, key, thr) ? v * scale : 0.f;
        x[e] = v;
    }
}

// thread = (token t, 4 columns) x a slice of the batch (blockIdx.y): dpos[t] += sum_b dz[b,t], dtok[t] likewise for the ns
// special tokens, dpatch[b, t-ns] = bf16(dz[b,t]) for t >= ns, with dz = dx * keep * scale.  The batch is sliced so that the
// grid fills the chip (one thread per (token, column quad) alone is ~150 waves); the slices meet in fp32 atomics on buffers
// the caller zeroes once per step.
__global__ void __launch_bounds__(256) embed_bwd_kernel(const float* __restrict__ dx, bf16_t* __restrict__ dpatch, float* __restrict__ dpos,
                                                        float* __restrict__ dcls, int B, int N, int D, int ns, float scale, uint32_t thr,
                                                        uint32_t key) {
    const int dq = D >> 2;
    const int64_t total = (int64_t)N * dq;
    const int per = (B + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx / dq), d = (int)(idx - (int64_t)t * dq) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int bb = b0; bb < b1; bb += 4) {     // four batch elements per trip: their loads are in flight together
            float4 v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = min(bb + u, b1 - 1);
                v4[u] = *reinterpret_cast<const float4*>(dx + ((int64_t)b * N + t) * D + d);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = bb + u;
                if (b >= b1) break;
                const int64_t e = ((int64_t)b * N + t) * D + d;
                float4 v = v4[u];
                if (thr) {
                    bool k0, k1, k2, k3;
                    chb_keep2((uint32_t)(e >> 1), key, thr, k0, k1);
                    chb_keep2((uint32_t)(e >> 1) + 1u, key, thr, k2, k3);
                    v.x = k0 ? v.x * scale : 0.f; v.y = k1 ? v.y * scale : 0.f;
                    v.z = k2 ? v.z * scale : 0.f; v.w = k3 ? v.w * scale : 0.f;
                }
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                if (t >= ns) {
                    uint2 o;
                    o.x = pack_bf16x2(v.x, v.y);
                    o.y = pack_bf16x2(v.z, v.w);
                    *reinterpret_cast<uint2*>(dpatch + ((int64_t)b * (N - ns) + (t - ns)) * D + d) = o;
                }
            }
        }
        float* pp = dpos + (int64_t)t * D + d;
        atomicAdd(pp + 0, acc.x); atomicAdd(pp + 1, acc.y); atomicAdd(pp + 2, acc.z); atomicAdd(pp + 3, acc.w);
        if (t < ns) {   // special tokens: class (row 0), distillation (row 1)
            float* pc = dcls + (int64_t)t * D + d;
            atomicAdd(pc + 0, acc.x); atomicAdd(pc + 1, acc.y); atomicAdd(pc + 2, acc.z); atomicAdd(pc + 3, acc.w);
        }
    }
}

__global__ void __launch_bounds__(256) dropout_bwd_kernel(const float* __restrict__ dy, int64_t ld, bf16_t* __restrict__ dz, int M, int N,
                                                          float scale, uint32_t thr, uint32_t key) {
    const int nq = N >> 2;
    const int64_t total = (int64_t)M * nq;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / nq;
        const int c = (int)(idx - r * nq) * 4;
        float4 v = *reinterpret_cast<const float4*>(dy + r * ld + c);
        if (thr) {
            const uint64_t e = (uint64_t)r * (uint64_t)N + (uint64_t)c;
            bool k0, k1, k2, k3;
            chb_keep2((uint32_t)(e >> 1), key, thr, k0, k1);
            chb_keep2((uint32_t)(e >> 1) + 1u, key, thr, k2, k3);
            v.x = k0 ? v.x * scale : 0.f; v.y = k1 ? v.y * scale : 0.f;
            v.z = k2 ? v.z * scale : 0.f; v.w = k3 ? v.w * scale : 0.f;
        }
        uint2 o;
        o.x = pack_bf16x2(v.x, v.y);
        o.y = pack_bf16x2(v.z, v.w);
        *reinterpret_cast<uint2*>(dz + r * (int64_t)N + c) = o;
    }
}

// grid = (col blocks of 256 columns, row slabs of 512 rows); lane owns 4 columns.
__global__ void __launch_bounds__(256) colsum_kernel(const bf16_t* __restrict__ x, int64_t ld, float* __restrict__ out, int M, int N) {
    __shared__ float red[4][256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = blockIdx.x * 256 + lane * 4;
    const int r0 = blockIdx.y * 512;
    int r1 = r0 + 512;
    r1 = r1 < M ? r1 : M;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < N) {
        for (int r = r0 + wave; r < r1; r += 4) {
            const uint2 v = *reinterpret_cast<const uint2*>(x + (int64_t)r * ld + c);
            acc.x += bf16_to_f32((bf16_t)(v.x & 0xffff)); acc.y += bf16_to_f32((bf16_t)(v.x >> 16));
            acc.z += bf16_to_f32((bf16_t)(v.y & 0xffff)); acc.w += bf16_to_f32((bf16_t)(v.y >> 16));
        }
    }
    red[wave][lane * 4 + 0] = acc.x; red[wave][lane * 4 + 1] = acc.y;
    red[wave][lane * 4 + 2] = acc.z; red[wave][lane * 4 + 3] = acc.w;
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < N) atomicAdd(out + cc, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// one wave per sample
__global__ void __launch_bounds__(256) softmax_ce_kernel(const float* __restrict__ logits, int64_t ld, const int32_t* __restrict__ labels,
                                                         float* __restrict__ loss, bf16_t* __restrict__ dl, int64_t ld_d, int B,
                                                         int classes, float grad_scale) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
        const float* z = logits + (int64_t)b * ld;
        float mx = -INFINITY;
        for (int c = lane; c < classes; c += 64) mx = fmaxf(mx, z[c]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int c = lane; c < classes; c += 64) s += expf(z[c] - mx);
        s = wave_sum(s);
        const int lab = labels[b];
        const bool lab_ok = (unsigned)lab < (unsigned)classes;    // labels are validated on the device: an out-of-range label (a -1
        const float lse = mx + logf(s);                           // "ignore" index, a corrupt value) never indexes the logits row;
        if (lane == 0) loss[b] = lab_ok ? lse - z[lab] : NAN;     // its loss is NaN (TF's sparse CE on CPU raises, on GPU yields NaN)
        if (dl) {                                                 // and its gradient row NaN, so the step cannot pass silently
            bf16_t* d = dl + (int64_t)b * ld_d;
            const float inv = 1.0f / s;
            for (int c = lane; c < (int)ld_d; c += 64) {
                float g = 0.f;
                if (c < classes) g = lab_ok ? (expf(z[c] - mx) * inv - (c == lab ? 1.0f : 0.0f)) * grad_scale : NAN;
                d[c] = f32_to_bf16(g);
            }
        }
    }
}

// grid = (tiles, matrices); 64x64 tile through LDS; desc = {src_off, dst_off, R, C}
__global__ void __launch_bounds__(256) cast_transpose_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, bf16_t* __restrict__ dst_t,
                                                             const int64_t* __restrict__ desc) {
    __shared__ bf16_t tile[64][66];
    const int64_t* d = desc + (int64_t)blockIdx.y * 4;
    const int64_t so = d[0], dof = d[1];
    const int R = (int)d[2], C = (int)d[3];
    const int tc = (C + 63) / 64, tr = (R + 63) / 64;
    if ((int)blockIdx.x >= tc * tr) return;
    const int r0 = ((int)blockIdx.x / tc) * 64, c0 = ((int)blockIdx.x % tc) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
        const int r = r0 + rr, c = c0 + tx;
        bf16_t v = 0;
        if (r < R && c < C) {
            v = f32_to_bf16(src[so + (int64_t)r * C + c]);
            if (dst) dst[dof + (int64_t)r * C + c] = v;
        }
        tile[rr][tx] = v;
    }
    if (!dst_t) return;
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        const int c = c0 + cc, r = r0 + tx;
        if (r < R && c < C) dst_t[dof + (int64_t)c * R + r] = tile[tx][cc];
    }
}

template <bool ZERO_G>
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    const uint8_t* __restrict__ flags, int64_t n4, float lr_t, float b1c, float b2c, float eps,
                                                    float wd, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        if (ZERO_G) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // the next backward accumulates into zeros
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        const bool decay = flags ? (flags[i >> 8] != 0) : true;  // 1024-element chunks = 256 float4
        float* pp = &pv.x; const float* gg = &gv.x; float* mm = &mv.x; float* vq = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] * gscale;
            float w = pp[k];
            if (decay) w = w - wd * w;                       // optimizers.py:147-155: decay first, wd not scaled by lr
            mm[k] = mm[k] + (gk - mm[k]) * b1c;              // keras Adam: m += (g - m)(1 - b1)
            vq[k] = vq[k] + (gk * gk - vq[k]) * b2c;
            pp[k] = w - lr_t * mm[k] / (sqrtf(vq[k]) + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
}

__global__ void __launch_bounds__(256) zero_f32_kernel(float* __restrict__ x, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        reinterpret_cast<float4*>(x)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- token pooling over the patch tokens 1..N-1 (pooling="avg" | "max" | "sum"); one thread = one (image, column pair)
__global__ void __launch_bounds__(256) pool_tokens_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ out, int32_t* __restrict__ argmax,
                                                          int B, int N, int D, int mode) {
    const int half = D >> 1;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * half) return;
    const int b = (int)(idx / half), c = (int)(idx - (int64_t)b * half) * 2;
    const bf16_t* p = h + (int64_t)b * N * D + c;
    float a0 = mode == CHB_POOL_MAX ? -INFINITY : 0.f, a1 = a0;
    int i0 = 1, i1 = 1;
    for (int t = 1; t < N; ++t) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(p + (int64_t)t * D);
        const float v0 = bf16_to_f32((bf16_t)(w & 0xffff)), v1 = bf16_to_f32((bf16_t)(w >> 16));
        if (mode == CHB_POOL_MAX) {
            if (v0 > a0) { a0 = v0; i0 = t; }
            if (v1 > a1) { a1 = v1; i1 = t; }
        } else {
            a0 += v0;
            a1 += v1;
        }
    }
    if (mode == CHB_POOL_AVG) {
        const float inv = 1.0f / (float)(N - 1);
        a0 *= inv;
        a1 *= inv;
    }
    *reinterpret_cast<uint32_t*>(out + (int64_t)b * D + c) = pack_bf16x2(a0, a1);
    if (mode == CHB_POOL_MAX && argmax) {
        argmax[(int64_t)b * D + c] = i0;
        argmax[(int64_t)b * D + c + 1] = i1;
    }
}

__global__ void __launch_bounds__(256) pool_tokens_bwd_kernel(const bf16_t* __restrict__ dout, const int32_t* __restrict__ argmax,
                                                              bf16_t* __restrict__ dh, int B, int N, int D, int mode) {
    const int half = D >> 1;
    const int64_t total = (int64_t)B * N * half;
    const float inv = 1.0f / (float)(N > 1 ? N - 1 : 1);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / half;
        const int c = (int)(idx - row * half) * 2;
        const int b = (int)(row / N), t = (int)(row - (int64_t)b * N);
        uint32_t w = 0;
        if (t > 0) {
            const uint32_t g = *reinterpret_cast<const uint32_t*>(dout + (int64_t)b * D + c);
            if (mode == CHB_POOL_SUM) {
                w = g;
            } else if (mode == CHB_POOL_AVG) {
                w = pack_bf16x2(bf16_to_f32((bf16_t)(g & 0xffff)) * inv, bf16_to_f32((bf16_t)(g >> 16)) * inv);
            } else {
                const int2 a = *reinterpret_cast<const int2*>(argmax + (int64_t)b * D + c);
                w = (a.x == t ? (g & 0xffffu) : 0u) | (a.y == t ? (g & 0xffff0000u) : 0u);
            }
        }
        *reinterpret_cast<uint32_t*>(dh + row * D + c) = w;
    }
}

// ---- tanh feature head
__global__ void __launch_bounds__(256) tanh_fwd_kernel(float* __restrict__ z, bf16_t* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = tanhf(z[i]);
        z[i] = v;
        if (y) y[i] = f32_to_bf16(v);
    }
}

__global__ void __launch_bounds__(256) tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, bf16_t* __restrict__ dz,
                                                       int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = y[i];
        dz[i] = f32_to_bf16(dy[i] * (1.0f - v * v));
    }
}

// ---- trainable stand-alone layers (round 3): the pieces torch.autograd.Function wrappers of chambers_amd/layers need that the
// whole-model engine has fused elsewhere - stand-alone GELU (activations.py:5-56) and its derivative, element-wise products for
// their backward, Dropout as a layer (flat element index, the chb_dropout_mask definition), the positional-table add with its
// batch reduction (layers/embedding.py:156-182) and the strided batch sum of ConcatEmbedding's backward (:218-261).
__global__ void __launch_bounds__(256) gelu_f32_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ d, int64_t n,
                                                       int approximate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        float yy, dd;
        if (approximate) {      // activations.py:30-44: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
            const float c = 0.7978845608028654f, k = 0.044715f;
            const float t = tanhf(c * (v + k * v * v * v));
            yy = 0.5f * v * (1.0f + t);
            dd = 0.5f * (1.0f + t) + 0.5f * v * (1.0f - t * t) * c * (1.0f + 3.0f * k * v * v);
        } else {
            gelu_both(v, yy, dd);
        }
        y[i] = yy;
        if (d) d[i] = dd;
    }
}

__global__ void __launch_bounds__(256) mul_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}

// out_bf16 = bf16(dy * aux): the backward of a Dense layer with the fused GELU epilogue (aux = the saved bf16 gelu'), one rounding -
// what the engine's gelu'-multiply GEMM epilogue does to its fp32 accumulator
template <bool DY_BF16>
__global__ void __launch_bounds__(256) scale_by_bf16_kernel(const void* __restrict__ dy, const bf16_t* __restrict__ aux, bf16_t* __restrict__ out,
                                                            int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = DY_BF16 ? bf16_to_f32(reinterpret_cast<const bf16_t*>(dy)[i]) : reinterpret_cast<const float*>(dy)[i];
        out[i] = f32_to_bf16(g * bf16_to_f32(aux[i]));
    }
}

__global__ void __launch_bounds__(256) dropout_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, float scale, uint32_t thr,
                                                          uint32_t key) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = chb_keep((uint64_t)i, key, thr) ? x[i] * scale : 0.0f;
}

__global__ void __launch_bounds__(256) add_rows_f32_kernel(const float* __restrict__ x, const float* __restrict__ table, float* __restrict__ out,
                                                           int64_t n, int64_t period) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = x[i] + table[i % period];
}

// out[c] = sum over r < rows of x[r * row_stride + c], c < cols (one thread per column: consecutive threads read consecutive addresses)
__global__ void __launch_bounds__(256) sum_rows_f32_kernel(const float* __restrict__ x, int64_t row_stride, int64_t rows, int64_t cols,
                                                           float* __restrict__ out) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += (int64_t)gridDim.x * blockDim.x) {
        float acc = 0.0f;
        for (int64_t r = 0; r < rows; ++r) acc += x[r * row_stride + c];
        out[c] = acc;
    }
}

}  // namespace

// ---- small tensor utilities of the stand-alone layers (Dense(softmax), ConcatEmbedding, EncoderLayer's first residual, operand casts):
// torch supplies memory, never arithmetic ----
__global__ void __launch_bounds__(256) add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n4,
                                                      int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = n4 * 4; i < n; ++i) out[i] = a[i] + b[i];
}

// out = alpha * a (+ beta * b): the average of the two heads of the distilled variant and its backward (vision_transformer.py:373-397)
__global__ void __launch_bounds__(256) axpby_f32_kernel(const float* __restrict__ a, float alpha, const float* __restrict__ b, float beta,
                                                        float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = b ? alpha * a[i] + beta * b[i] : alpha * a[i];
}

// x[r][c] = bf16(x[r][c] + y[r][c]) over strided bf16 rows (fp32 sum, one rounding)
__global__ void __launch_bounds__(256) add_rows_bf16_kernel(bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ y, int64_t ldy, int64_t rows,
                                                            int cols) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        x[r * ldx + c] = f32_to_bf16(bf16_to_f32(x[r * ldx + c]) + bf16_to_f32(y[r * ldy + c]));
    }
}

__global__ void __launch_bounds__(256) cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = f32_to_bf16(src[i]);
}

__global__ void __launch_bounds__(256) cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = bf16_to_f32(src[i]);
}

// rows x row_bytes, independent strides (bytes): dst[r][0..row_bytes) = src[r][0..row_bytes); one wave per row chunk of 256 bytes
__global__ void __launch_bounds__(256) copy_rows_kernel(const uint8_t* __restrict__ src, int64_t src_stride, uint8_t* __restrict__ dst,
                                                        int64_t dst_stride, int64_t rows, int64_t row_bytes) {
    const int64_t chunks = (row_bytes + 3) / 4;       // dword granules (row_bytes % 4 == 0 and 4-byte alignment are checked on the host)
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks, c = i - r * chunks;
        reinterpret_cast<uint32_t*>(dst + r * dst_stride)[c] = reinterpret_cast<const uint32_t*>(src + r * src_stride)[c];
    }
}

// softmax over the last axis, one wave per row, fp32 (tf.nn.softmax: exp(x - max) / sum)
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* __restrict__ x, int64_t ld, float* __restrict__ out, int64_t ld_out, int rows,
                                                           int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * ld;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, xr[c]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.0f;
    for (int c = lane; c < cols; c += 64) sum += expf(xr[c] - m);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    float* orow = out + (int64_t)row * ld_out;
    for (int c = lane; c < cols; c += 64) orow[c] = expf(xr[c] - m) / sum;
}

int chb_option(int id) {
    if (id < 0 || id >= CHB_OPT_COUNT) return 0;
    if (!g_option_read[id].load(std::memory_order_acquire)) {
        char env[64] = "CHB_";
        strncat(env, kOptions[id].name, sizeof(env) - 5);
        const char* e = getenv(env);
        g_option[id].store(e ? atoi(e) : kOptions[id].dflt, std::memory_order_relaxed);
        g_option_read[id].store(true, std::memory_order_release);
    }
    return g_option[id].load(std::memory_order_relaxed);
}

extern "C" {

int chb_set_option(const char* name, int value) {
    if (!name) return CHB_EINVAL;
    for (int id = 0; id < CHB_OPT_COUNT; ++id)
        if (!strcmp(name, kOptions[id].name)) {
            g_option[id].store(value, std::memory_order_relaxed);
            g_option_read[id].store(true, std::memory_order_release);
            return CHB_OK;
        }
    return CHB_EINVAL;
}

int chb_version(void) { return 4; }
const char* chb_build_arch(void) { return "gfx950"; }

int chb_dropout_mask(uint8_t* out, int64_t n, float rate, uint32_t key, void* stream) {
    if (!out || n < 0 || rate < 0.f || rate >= 1.f) return CHB_EINVAL;
    if (n == 0) return CHB_OK;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out, n,
                       rate > 0.f ? chb_drop_threshold(rate) : 0u, key);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_token_row(float* x, const float* tok, const float* pos, int B, int N, int D, int row, float drop_rate, uint32_t drop_key,
                  void* stream) {
    if (!x || !tok || !pos || B < 0 || N <= 0 || D <= 0 || row < 0 || row >= N || drop_rate < 0.f || drop_rate >= 1.f) return CHB_EINVAL;
    if (B == 0) return CHB_OK;
    hipLaunchKernelGGL(cls_row_kernel, dim3(grid_for((int64_t)B * D)), dim3(256), 0, (hipStream_t)stream, x, tok, pos, B, N, D, row,
                       1.0f / (1.0f - drop_rate), drop_rate > 0.f ? chb_drop_threshold(drop_rate) : 0u, drop_key);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_cls_row(float* x, const float* cls, const float* pos, int B, int N, int D, float drop_rate, uint32_t drop_key, void* stream) {
    return chb_token_row(x, cls, pos, B, N, D, 0, drop_rate, drop_key, stream);
}

int chb_embed_bwd_tokens(const float* dx, void* dpatch, float* dpos, float* dtok, int B, int N, int D, int n_special, float drop_rate,
                         uint32_t drop_key, void* stream) {
    if (!dx || !dpatch || !dpos || !dtok || B < 0 || n_special < 1 || N <= n_special || D <= 0 || (D & 3) || drop_rate < 0.f ||
        drop_rate >= 1.f)
        return CHB_EINVAL;
    if (B == 0) return CHB_OK;
    const int gx = grid_for((int64_t)N * (D / 4));
    int slices = 2048 / gx;                       // ~2048 workgroups in total
    slices = slices < 1 ? 1 : (slices > B ? B : slices);
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(gx, slices), dim3(256), 0, (hipStream_t)stream, dx, (bf16_t*)dpatch,
                       dpos, dtok, B, N, D, n_special, 1.0f / (1.0f - drop_rate), drop_rate > 0.f ? chb_drop_threshold(drop_rate) : 0u,
                       drop_key);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_embed_bwd(const float* dx, void* dpatch, float* dpos, float* dcls, int B, int N, int D, float drop_rate, uint32_t drop_key,
                  void* stream) {
    return chb_embed_bwd_tokens(dx, dpatch, dpos, dcls, B, N, D, 1, drop_rate, drop_key, stream);
}

int chb_dropout_bwd_bf16(const float* dy, int64_t ld, void* dz, int M, int N, float drop_rate, uint32_t drop_key, void* stream) {
    if (!dy || !dz || M < 0 || N <= 0 || (N & 3) || (ld & 3) || drop_rate < 0.f || drop_rate >= 1.f) return CHB_EINVAL;
    if (M == 0) return CHB_OK;
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for((int64_t)M * (N / 4))), dim3(256), 0, (hipStream_t)stream, dy, ld, (bf16_t*)dz, M,
                       N, 1.0f / (1.0f - drop_rate), drop_rate > 0.f ? chb_drop_threshold(drop_rate) : 0u, drop_key);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_colsum_bf16(const void* x, int64_t ld, float* out, int M, int N, void* stream) {
    if (!x || !out || M < 0 || N <= 0 || (N & 3) || (ld & 3)) return CHB_EINVAL;
    if (M == 0) return CHB_OK;
    hipLaunchKernelGGL(colsum_kernel, dim3(chb_div_up(N, 256), chb_div_up(M, 512)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld,
                       out, M, N);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_pool_tokens(const void* h, void* out, int32_t* argmax, int B, int N, int D, int mode, void* stream) {
    if (B < 0 || N < 2 || D <= 0 || (D & 1) || mode < CHB_POOL_AVG || mode > CHB_POOL_SUM) return CHB_EINVAL;
    if (B == 0) return CHB_OK;
    if (!h || !out || (mode == CHB_POOL_MAX && !argmax)) return CHB_EINVAL;
    hipLaunchKernelGGL(pool_tokens_kernel, dim3(chb_div_up((int64_t)B * (D / 2), 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h,
                       (bf16_t*)out, argmax, B, N, D, mode);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_pool_tokens_bwd(const void* dout, const int32_t* argmax, void* dh, int B, int N, int D, int mode, void* stream) {
    if (B < 0 || N < 2 || D <= 0 || (D & 1) || mode < CHB_POOL_AVG || mode > CHB_POOL_SUM) return CHB_EINVAL;
    if (B == 0) return CHB_OK;
    if (!dout || !dh || (mode == CHB_POOL_MAX && !argmax)) return CHB_EINVAL;
    hipLaunchKernelGGL(pool_tokens_bwd_kernel, dim3(grid_for((int64_t)B * N * (D / 2))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dout, argmax, (bf16_t*)dh, B, N, D, mode);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_tanh_fwd(float* z, void* y_bf16, int64_t n, void* stream) {
    if (n < 0) return CHB_EINVAL;
    if (n == 0) return CHB_OK;
    if (!z) return CHB_EINVAL;
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, z, (bf16_t*)y_bf16, n);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_tanh_bwd(const float* dy, const float* y, void* dz_bf16, int64_t n, void* stream) {
    if (n < 0) return CHB_EINVAL;
    if (n == 0) return CHB_OK;
    if (!dy || !y || !dz_bf16) return CHB_EINVAL;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, y, (bf16_t*)dz_bf16, n);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_softmax_ce(const float* logits, int64_t ld, const int32_t* labels, float* loss_per_sample, void* dlogits, int64_t ld_d, int B,
                   int classes, float grad_scale, void* stream) {
    if (!logits || !labels || !loss_per_sample || B < 0 || classes <= 0 || ld < classes) return CHB_EINVAL;
    if (dlogits && ld_d < classes) return CHB_EINVAL;
    if (B == 0) return CHB_OK;
    int blocks = chb_div_up(B, 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, loss_per_sample,
                       (bf16_t*)dlogits, ld_d, B, classes, grad_scale);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_cast_transpose(const float* src, void* dst, void* dst_t, const int64_t* desc, int n_desc, int max_tiles, void* stream) {
    if (!src || !desc || n_desc < 0 || max_tiles <= 0 || (!dst && !dst_t)) return CHB_EINVAL;
    if (n_desc == 0) return CHB_OK;
    hipLaunchKernelGGL(cast_transpose_kernel, dim3(max_tiles, n_desc), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst,
                       (bf16_t*)dst_t, desc);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_adamw(float* p, float* g, float* m, float* v, const uint8_t* decay_flags, int64_t n, float lr_t, float beta1, float beta2,
              float eps, float weight_decay, float grad_scale, int zero_grad, void* stream) {
    if (!p || !g || !m || !v || n < 0 || (n & 3)) return CHB_EINVAL;
    if (n == 0) return CHB_OK;
    if (zero_grad)
        hipLaunchKernelGGL(adamw_kernel<true>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, decay_flags, n / 4, lr_t,
                           1.0f - beta1, 1.0f - beta2, eps, weight_decay, grad_scale);
    else
        hipLaunchKernelGGL(adamw_kernel<false>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, decay_flags, n / 4, lr_t,
                           1.0f - beta1, 1.0f - beta2, eps, weight_decay, grad_scale);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (n == 0) return CHB_OK;
    if (!a || !b || !out || n < 0) return CHB_EINVAL;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) return CHB_EINVAL;
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4, n);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_axpby_f32(const float* a, float alpha, const float* b, float beta, float* out, int64_t n, void* stream) {
    if (n == 0) return CHB_OK;
    if (!a || !out || n < 0) return CHB_EINVAL;
    hipLaunchKernelGGL(axpby_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, alpha, b, beta, out, n);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_add_rows_bf16(void* x_bf16, int64_t ldx, const void* y_bf16, int64_t ldy, int64_t rows, int cols, void* stream) {
    if (rows == 0 || cols == 0) return CHB_OK;
    if (!x_bf16 || !y_bf16 || rows < 0 || cols < 0 || ldx < cols || ldy < cols) return CHB_EINVAL;
    hipLaunchKernelGGL(add_rows_bf16_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x_bf16, ldx, (const bf16_t*)y_bf16, ldy,
                       rows, cols);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
    if (n == 0) return CHB_OK;
    if (!src || !dst || n < 0) return CHB_EINVAL;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_cast_bf16_f32(const void* src, float* dst, int64_t n, void* stream) {
    if (n == 0) return CHB_OK;
    if (!src || !dst || n < 0) return CHB_EINVAL;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, n);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_copy_rows(const void* src, int64_t src_stride_bytes, void* dst, int64_t dst_stride_bytes, int64_t rows, int64_t row_bytes, void* stream) {
    if (rows == 0 || row_bytes == 0) return CHB_OK;
    if (!src || !dst || rows < 0 || row_bytes < 0) return CHB_EINVAL;
    if ((row_bytes & 3) || (src_stride_bytes & 3) || (dst_stride_bytes & 3) || ((uintptr_t)src & 3) || ((uintptr_t)dst & 3)) return CHB_EINVAL;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(rows * (row_bytes / 4))), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src,
                       src_stride_bytes, (uint8_t*)dst, dst_stride_bytes, rows, row_bytes);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_softmax_f32(const float* x, int64_t ld, float* out, int64_t ld_out, int rows, int cols, void* stream) {
    if (rows == 0) return CHB_OK;
    if (!x || !out || rows < 0 || cols <= 0 || ld < cols || ld_out < cols) return CHB_EINVAL;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(chb_div_up(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ld, out, ld_out, rows, cols);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_zero_f32(float* x, int64_t n, void* stream) {
    if (n < 0 || (n && !x)) return CHB_EINVAL;
    if (n == 0) return CHB_OK;
    if (((uintptr_t)x & 15) || (n & 3)) return CHB_EINVAL;
    hipLaunchKernelGGL(zero_f32_kernel, dim3(grid_for(n / 4, 4096)), dim3(256), 0, (hipStream_t)stream, x, n / 4);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_gelu_f32(const float* x, float* y, float* dydx, int64_t n, int approximate, void* stream) {
    if (n == 0) return CHB_OK;
    if (!x || !y || n < 0) return CHB_EINVAL;
    hipLaunchKernelGGL(gelu_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, dydx, n, approximate);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_mul_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (n == 0) return CHB_OK;
    if (!a || !b || !out || n < 0) return CHB_EINVAL;
    hipLaunchKernelGGL(mul_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_scale_by_bf16(const void* dy, int dy_dtype, const void* aux_bf16, void* out_bf16, int64_t n, void* stream) {
    if (n == 0) return CHB_OK;
    if (!dy || !aux_bf16 || !out_bf16 || n < 0) return CHB_EINVAL;
    if (dy_dtype == CHB_OUT_BF16)
        hipLaunchKernelGGL(scale_by_bf16_kernel<true>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, (const bf16_t*)aux_bf16, (bf16_t*)out_bf16, n);
    else if (dy_dtype == CHB_OUT_F32)
        hipLaunchKernelGGL(scale_by_bf16_kernel<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, (const bf16_t*)aux_bf16, (bf16_t*)out_bf16, n);
    else
        return CHB_EINVAL;
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_dropout_f32(const float* x, float* out, int64_t n, float rate, uint32_t key, void* stream) {
    if (n == 0) return CHB_OK;
    if (!x || !out || n < 0 || rate < 0.0f || rate >= 1.0f) return CHB_EINVAL;
    hipLaunchKernelGGL(dropout_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, out, n, 1.0f / (1.0f - rate),
                       rate > 0.0f ? chb_drop_threshold(rate) : 0u, key);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_add_rows_f32(const float* x, const float* table, float* out, int64_t n, int64_t period, void* stream) {
    if (n == 0) return CHB_OK;
    if (!x || !table || !out || n < 0 || period <= 0) return CHB_EINVAL;
    hipLaunchKernelGGL(add_rows_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, table, out, n, period);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_sum_rows_f32(const float* x, int64_t row_stride, int64_t rows, int64_t cols, float* out, void* stream) {
    if (cols == 0) return CHB_OK;
    if (!x || !out || rows < 0 || cols < 0 || row_stride < cols) return CHB_EINVAL;
    hipLaunchKernelGGL(sum_rows_f32_kernel, dim3(grid_for(cols)), dim3(256), 0, (hipStream_t)stream, x, row_stride, rows, cols, out);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

}  // extern "C"
