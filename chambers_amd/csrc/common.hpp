// Shared device helpers for the chambers MI355X (gfx950) kernels.
// Wavefront = 64 lanes everywhere in this tree; no 32-wide idioms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CHB_OK 0
#define CHB_EINVAL (-1)
#define CHB_ELAUNCH (-2)
#define CHB_EUNSUPPORTED (-3)

#define CHB_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return CHB_ELAUNCH;           \
    } while (0)

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef __attribute__((ext_vector_type(16))) float float16_t;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}

// round-to-nearest-even; a plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// ---- counter-hash RNG for dropout (definition: oracle/rng_ref.py) -------------
__device__ __forceinline__ uint32_t chb_hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// 16-bit uniform of flat element index e under site key `key`
__device__ __forceinline__ uint32_t chb_u16(uint64_t e, uint32_t key) {
    uint32_t r = chb_hash32(((uint32_t)(e >> 1)) ^ key);
    return (r >> (16u * (uint32_t)(e & 1))) & 0xffffu;
}

__device__ __forceinline__ bool chb_keep(uint64_t e, uint32_t key, uint32_t thr) {
    return chb_u16(e, key) >= thr;
}

// both halves of one hash: elements 2c and 2c+1
__device__ __forceinline__ void chb_keep2(uint32_t c, uint32_t key, uint32_t thr, bool& k0, bool& k1) {
    uint32_t r = chb_hash32(c ^ key);
    k0 = (r & 0xffffu) >= thr;
    k1 = (r >> 16) >= thr;
}

__host__ __device__ __forceinline__ uint32_t chb_drop_threshold(float rate) {
    return (uint32_t)(rate * 65536.0f + 0.5f);
}

// ---- wave / block reductions ----------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact-erf GELU and its derivative (activations.py:46-56).  erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e.
// fp32-level): one v_rcp, one v_exp and five FMAs; the exponential exp(-x^2/2) is shared with the Gaussian density the
// derivative needs.
// both at once (shared erf / exp): y = gelu(x), d = gelu'(x).  Written on the normal CDF directly: with h = 0.5 * erfc(|x| / sqrt 2)
// (the A&S polynomial with halved coefficients), cdf = 1 - h for x >= 0 and h otherwise; 1 + p z is one FMA on |x|.
__device__ __forceinline__ void gelu_both(float x, float& y, float& d) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
    const float ex = __builtin_amdgcn_exp2f((x * x) * (-0.5f * 1.44269504088896340736f));     // exp(-x^2 / 2)
    float poly = 0.5f * 1.061405429f;
    poly = poly * t - 0.5f * 1.453152027f;
    poly = poly * t + 0.5f * 1.421413741f;
    poly = poly * t - 0.5f * 0.284496736f;
    poly = poly * t + 0.5f * 0.254829592f;
    const float h = (poly * t) * ex;
    const float cdf = x >= 0.0f ? 1.0f - h : h;
    y = x * cdf;
    d = cdf + (x * 0.39894228040143267794f) * ex;
}

static inline int chb_div_up(long a, long b) { return (int)((a + b - 1) / b); }

// ---- host-side tuning / A-B switches --------------------------------------------------------
// Each switch takes its default from the environment variable CHB_<NAME> ONCE per process (first use) and can be changed at run
// time through chb_set_option (tests and tools/ cross-check two algorithms in one process): no getenv on the launch path.
enum ChbOption {
    CHB_OPT_ATTN_FWD_ALGO = 0,   // 0 auto (persistent pipelined kernel for 193 <= N <= 208, whole-head kernel for other N <= 224), 1 resident (N <= 224), 2 streaming, 3 whole-head kernel also for 193..208 (A/B)
    CHB_OPT_ATTN_BWD_ALGO,       // 0 auto (persistent pipelined kernel for 193 <= N <= 208, else the lean one-pass kernel; the 16-wave kernel when the fused bias gradient is asked for), 1 resident 8 waves, 2 two-pass, 3 resident 16 waves, 4 lean kernel also for 193..208 (A/B), 5 = 0
    CHB_OPT_AFFINE_ALGO,         // 0 auto, 1 rows, 2 32x8 tiles, 3 16x16 tiles
    CHB_OPT_GEMM_ALGO,           // 0 auto, 1 128x128 tiles, 2 persistent 256x256, 3 persistent 128x256 x 2 workgroups / CU, 4 persistent 256x256 ping-pong (full tiles), 5 persistent 256x256 with software-pipelined fragment reads
    CHB_OPT_GEMM_WALK,           // 0 linear tile ids per XCD, 1 (default) panel walk where it pays, 2 always panel
    CHB_OPT_TN_ATOMICS,          // 1 = always the atomic epilogue
    CHB_OPT_TN_FAST,             // 0 = generic staging addresses everywhere (default 1)
    CHB_OPT_GEMM_TILE_QUEUE,     // 1 = persistent NT GEMM workgroups claim their tiles after the first from per-XCD counters (the engine turns it on when it trains data-parallel), 0 (default) = static tile shares
    CHB_OPT_LN_STREAM,           // LayerNorm: 1 = x / dy / the old dx are read with non-temporal loads (they are not read again before they leave the caches), 0 = plain loads
    CHB_OPT_DEBUG,               // timing experiments only (tools/): e.g. 1 = attention backward without its main loop; results are WRONG
    CHB_OPT_COUNT
};
int chb_option(int id);   // defined in elementwise.hip

// ---- launch profiler (csrc/vit_block.hip; chb_profile_enable / chb_profile_collect in the header): a slot, or -1 when recording is off
int chb_prof_begin(int kind, int epi, int out_dtype, int64_t m, int64_t n, int64_t k, hipStream_t s);
void chb_prof_end(int slot, int family, hipStream_t s);
