// General scaled dot-product attention: the parts of the ScaledAttention / MultiHeadAttention signature the ViT never uses -
// value (key-padding) mask, query mask, causal mask, cross-attention with Tq != Tk, any head_dim <= 128
// (layers/attention.py:7-23 ScaledAttention on keras Attention, :99-153 MultiHeadAttention.call / separate_heads_mask).
// keras BaseDenseAttention._apply_scores: scores -= 1e9 * (1 - mask) with mask = value_mask AND causal lower triangle; softmax;
// dropout on the weights; weights . value; result *= query_mask.
//
// NOT the hot path (the ViT block goes through csrc/attention.hip: MFMA, LDS-resident heads).  This file is the functional
// completion of the layer API: one wave per query row, scores of the row in LDS, fp32 arithmetic throughout, plain loads (K and V of
// a head are re-read from L2 by every query row of the head).  The dropout keep mask uses the same counter hash and the same
// element index convention as the fused kernels: ((b*H + h)*Tq + q) * Tk4 + k with Tk4 = Tk rounded up to a multiple of 4.
#include "common.hpp"
#include "../../include/chambers_hip.h"
#include <math.h>
#include <atomic>

namespace {

constexpr int GA_MAX_TK = 4096;     // scores of one query row per wave in LDS: 4 waves x 16 KiB
constexpr int GA_MAX_HD = 128;

struct GaParams {
    const bf16_t* q; int64_t ldq;
    const bf16_t* k; int64_t ldk;
    const bf16_t* v; int64_t ldv;
    int B, Tq, Tk, H, hd;
    const uint8_t* vmask;     // [B, Tk] or null
    const uint8_t* qmask;     // [B, Tq] or null
    int causal;
    float scale, drop_scale;
    uint32_t drop_thr, drop_key;
};

__device__ __forceinline__ bool ga_allowed(const GaParams& p, int b, int i, int j) {
    if (p.vmask && !p.vmask[(int64_t)b * p.Tk + j]) return false;
    if (p.causal && j > i) return false;
    return true;
}

// masked score row of query (b, h, i) into s[0..Tk) (fp32), returns max and sum of exp(s - max) through references
__device__ __forceinline__ void ga_scores(const GaParams& p, int b, int h, int i, int lane, const float* qrow, float* s, float& mx, float& sum) {
    mx = -INFINITY;
    for (int j = lane; j < p.Tk; j += 64) {
        const bf16_t* kr = p.k + ((int64_t)b * p.Tk + j) * p.ldk + h * p.hd;
        float acc = 0.f;
        for (int d = 0; d < p.hd; d += 2) {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(kr + d);
            acc = __builtin_fmaf(qrow[d], bf16_to_f32((bf16_t)(w & 0xffff)), acc);
            acc = __builtin_fmaf(qrow[d + 1], bf16_to_f32((bf16_t)(w >> 16)), acc);
        }
        float sc = acc * p.scale;
        if (!ga_allowed(p, b, i, j)) sc -= 1e9f;          // keras: scores -= 1e9 * cast(logical_not(mask))
        s[j] = sc;
        mx = fmaxf(mx, sc);
    }
    mx = wave_max(mx);
    sum = 0.f;
    for (int j = lane; j < p.Tk; j += 64) sum += __expf(s[j] - mx);
    sum = wave_sum(sum);
}

__device__ __forceinline__ float ga_keep(const GaParams& p, int b, int h, int i, int j) {
    if (!p.drop_thr) return 1.0f;
    const uint32_t tk4 = (uint32_t)((p.Tk + 3) & ~3);
    const uint64_t e = ((uint64_t)((uint32_t)(b * p.H + h)) * (uint32_t)p.Tq + (uint32_t)i) * tk4 + (uint32_t)j;
    return chb_keep(e, p.drop_key, p.drop_thr) ? p.drop_scale : 0.0f;
}

__global__ void __launch_bounds__(256) ga_fwd_kernel(GaParams p, bf16_t* __restrict__ o, int64_t ldo, float* __restrict__ lse) {
    extern __shared__ float ga_smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int i = blockIdx.y * 4 + wave;
    float* qrow = ga_smem + wave * (GA_MAX_HD + p.Tk);
    float* s = qrow + GA_MAX_HD;
    if (i >= p.Tq) return;                       // whole wave; no workgroup barrier below
    const bf16_t* qr = p.q + ((int64_t)b * p.Tq + i) * p.ldq + h * p.hd;
    for (int d = lane; d < p.hd; d += 64) qrow[d] = bf16_to_f32(qr[d]);
    __builtin_amdgcn_wave_barrier();
    float mx, sum;
    ga_scores(p, b, h, i, lane, qrow, s, mx, sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < p.Tk; j += 64) s[j] = __expf(s[j] - mx) * inv * ga_keep(p, b, h, i, j);     // dropped weights
    __builtin_amdgcn_wave_barrier();
    const float qm = (p.qmask && !p.qmask[(int64_t)b * p.Tq + i]) ? 0.0f : 1.0f;
    for (int d = lane; d < p.hd; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < p.Tk; ++j) acc = __builtin_fmaf(s[j], bf16_to_f32(p.v[((int64_t)b * p.Tk + j) * p.ldv + h * p.hd + d]), acc);
        o[((int64_t)b * p.Tq + i) * ldo + h * p.hd + d] = f32_to_bf16(acc * qm);
    }
    if (lane == 0) lse[(int64_t)bh * p.Tq + i] = mx + __logf(sum);
}

// backward of one query row: dq written, dk / dv accumulated with fp32 atomics into caller-zeroed buffers [B, Tk, H*hd]
__global__ void __launch_bounds__(256) ga_bwd_kernel(GaParams p, const bf16_t* __restrict__ o, int64_t ldo, const bf16_t* __restrict__ d_o,
                                                     int64_t ldg, const float* __restrict__ lse, float* __restrict__ dq, float* __restrict__ dk,
                                                     float* __restrict__ dv) {
    extern __shared__ float ga_smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int i = blockIdx.y * 4 + wave;
    float* qrow = ga_smem + wave * (2 * GA_MAX_HD + 2 * p.Tk);
    float* grow = qrow + GA_MAX_HD;
    float* s = grow + GA_MAX_HD;        // probabilities, then dS
    float* pd = s + p.Tk;               // dropped probabilities
    if (i >= p.Tq) return;
    const int D = p.H * p.hd;
    const float qm = (p.qmask && !p.qmask[(int64_t)b * p.Tq + i]) ? 0.0f : 1.0f;
    const bf16_t* qr = p.q + ((int64_t)b * p.Tq + i) * p.ldq + h * p.hd;
    const bf16_t* gr = d_o + ((int64_t)b * p.Tq + i) * ldg + h * p.hd;
    const bf16_t* orow = o + ((int64_t)b * p.Tq + i) * ldo + h * p.hd;
    float delta = 0.f;
    for (int d = lane; d < p.hd; d += 64) {
        qrow[d] = bf16_to_f32(qr[d]);
        const float g = bf16_to_f32(gr[d]) * qm;            // result *= query_mask: the gradient into the weighted sum
        grow[d] = g;
        delta += g * bf16_to_f32(orow[d]);                  // o already carries the query mask; a masked row has g = 0
    }
    delta = wave_sum(delta);
    __builtin_amdgcn_wave_barrier();
    float mx, sum;
    ga_scores(p, b, h, i, lane, qrow, s, mx, sum);
    const float l = lse[(int64_t)bh * p.Tq + i];
    for (int j = lane; j < p.Tk; j += 64) {
        const float pr = __expf(s[j] - l);
        const float keepc = ga_keep(p, b, h, i, j);
        const bf16_t* vr = p.v + ((int64_t)b * p.Tk + j) * p.ldv + h * p.hd;
        float dp = 0.f;
        for (int d = 0; d < p.hd; ++d) dp = __builtin_fmaf(grow[d], bf16_to_f32(vr[d]), dp);
        pd[j] = pr * keepc;
        s[j] = pr * (dp * keepc - delta) * p.scale;         // d(raw score . scale): the scale folded in once
    }
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < p.hd; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < p.Tk; ++j) acc = __builtin_fmaf(s[j], bf16_to_f32(p.k[((int64_t)b * p.Tk + j) * p.ldk + h * p.hd + d]), acc);
        dq[((int64_t)b * p.Tq + i) * D + h * p.hd + d] = acc;
    }
    for (int j = 0; j < p.Tk; ++j) {
        const float dsj = s[j], pdj = pd[j];
        if (dsj == 0.f && pdj == 0.f) continue;             // wave-uniform (LDS values): masked / dropped keys cost nothing
        for (int d = lane; d < p.hd; d += 64) {
            const int64_t at = ((int64_t)b * p.Tk + j) * D + h * p.hd + d;
            atomicAdd(dk + at, dsj * qrow[d]);
            atomicAdd(dv + at, pdj * grow[d]);
        }
    }
}

int ga_check(const void* q, const void* k, const void* v, int B, int Tq, int Tk, int H, int hd, float drop_rate) {
    if (!q || !k || !v || B < 0 || Tq <= 0 || Tk <= 0 || H <= 0 || hd <= 0 || drop_rate < 0.f || drop_rate >= 1.f) return CHB_EINVAL;
    if (hd > GA_MAX_HD || (hd & 1) || Tk > GA_MAX_TK) return CHB_EUNSUPPORTED;
    if (drop_rate > 0.f && (double)B * H * Tq * ((Tk + 3) & ~3) >= 4294967296.0) return CHB_EUNSUPPORTED;   // 32-bit dropout element index
    return CHB_OK;
}

}  // namespace

extern "C" {

int chb_attention_general_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo,
                              float* lse, int B, int Tq, int Tk, int H, int hd, const uint8_t* value_mask, const uint8_t* query_mask,
                              int causal, float drop_rate, uint32_t drop_key, float scale, void* stream) {
    const int rc = ga_check(q, k, v, B, Tq, Tk, H, hd, drop_rate);
    if (rc != CHB_OK) return rc;
    if (!o || !lse || (ldq & 1) || (ldk & 1)) return CHB_EINVAL;
    if (B == 0) return CHB_OK;
    GaParams p{(const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, B, Tq, Tk, H, hd, value_mask, query_mask, causal,
               scale > 0.f ? scale : 1.0f / sqrtf((float)hd), 1.0f / (1.0f - drop_rate), drop_rate > 0.f ? chb_drop_threshold(drop_rate) : 0u, drop_key};
    const size_t lds = (size_t)4 * (GA_MAX_HD + Tk) * sizeof(float);
    static std::atomic<bool> attr_fwd[64];       // once per DEVICE (the attribute is per device): a driver call, not per launch
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return CHB_ELAUNCH;
    std::atomic<bool>& attr = attr_fwd[dev];
    if (!attr.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)ga_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (GA_MAX_HD + GA_MAX_TK) * 4) != hipSuccess)
            return CHB_ELAUNCH;
        attr.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(ga_fwd_kernel, dim3(B * H, chb_div_up(Tq, 4)), dim3(256), lds, (hipStream_t)stream, p, (bf16_t*)o, ldo, lse);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_attention_general_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o, int64_t ldo,
                              const void* d_o, int64_t ldg, const float* lse, float* dq, float* dk, float* dv, int B, int Tq, int Tk, int H,
                              int hd, const uint8_t* value_mask, const uint8_t* query_mask, int causal, float drop_rate, uint32_t drop_key,
                              float scale, void* stream) {
    const int rc = ga_check(q, k, v, B, Tq, Tk, H, hd, drop_rate);
    if (rc != CHB_OK) return rc;
    if (!o || !d_o || !lse || !dq || !dk || !dv || (ldq & 1) || (ldk & 1)) return CHB_EINVAL;
    if (B == 0) return CHB_OK;
    GaParams p{(const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, B, Tq, Tk, H, hd, value_mask, query_mask, causal,
               scale > 0.f ? scale : 1.0f / sqrtf((float)hd), 1.0f / (1.0f - drop_rate), drop_rate > 0.f ? chb_drop_threshold(drop_rate) : 0u, drop_key};
    const size_t lds = (size_t)4 * (2 * GA_MAX_HD + 2 * Tk) * sizeof(float);
    static std::atomic<bool> attr_bwd[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return CHB_ELAUNCH;
    std::atomic<bool>& attr = attr_bwd[dev];
    if (!attr.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)ga_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (2 * GA_MAX_HD + 2 * GA_MAX_TK) * 4) != hipSuccess)
            return CHB_ELAUNCH;
        attr.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(ga_bwd_kernel, dim3(B * H, chb_div_up(Tq, 4)), dim3(256), lds, (hipStream_t)stream, p, (const bf16_t*)o, ldo,
                       (const bf16_t*)d_o, ldg, lse, dq, dk, dv);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

}  // extern "C"
