// Fused multi-head attention forward/backward for ViT sequence lengths (N = 197 / 577), gfx950.
//
// One workgroup (4 waves) owns one (image, head).  N is small enough that the head's K and V
// (and in the backward Q and dO too) stay resident in LDS, so scores never touch HBM:
//   forward : S^T = K.Q^T (keys on MFMA rows -> a lane holds ONE query's scores, softmax is an
//             in-lane reduction + 2 shuffles), P^T feeds the P.V MFMA straight from the
//             accumulators (k-order permuted on both operands), V^T via ds_read_b64_tr_b16.
//   backward: each of 8 waves owns 1/8 of the key tiles and keeps dK^T/dV^T for them in registers
//             across all query blocks; dS goes once through LDS for dQ.  No atomics.  N > 224 takes
//             the two-pass kernels further down (dK/dV per key chunk, dQ per query chunk), and the
//             streaming forward with online softmax.
// Dropout on the probabilities uses the counter hash of common.hpp, element index
// ((b*H + h)*N + q)*N + k, so forward and backward regenerate the same mask.
//
// Replaces keras Attention under chambers' ScaledAttention / MultiHeadAttention:
// layers/attention.py:13-23 (scores / sqrt(head_dim) after the matmul), :120-125.
#include "common.hpp"
#include "../../include/chambers_hip.h"
#include <atomic>

namespace {

constexpr int HD = 64;  // head dim (all ViT configs of the reference use 64)

__device__ __forceinline__ int swz_row(int r) { return (r >> 1) & 7; }          // row reads (ds_read_b128)
__device__ __forceinline__ int swz_trv(int r) { return ((r >> 1) & 3) << 1; }   // V image, transposed reads only

// row fragment: 8 consecutive d of row r, chunk index `chunk` (0..7), image swizzled with swz_row
__device__ __forceinline__ bf16x8_t lds_row_frag(const bf16_t* img, int r, int chunk) {
    return *reinterpret_cast<const bf16x8_t*>(img + r * HD + ((chunk ^ swz_row(r)) << 3));
}

// transposed fragment for MFMA lane (g, i): element j<4 = img[ra + j'][c0 + i], j>=4 = img[rb + j'][c0 + i],
// where this lane supplies the addresses of rows ra + (i>>2) / rb + (i>>2), columns c0 + 4*(i&3).
template <bool VSWZ>
__device__ __forceinline__ bf16x8_t lds_tr_frag(const bf16_t* img, int ra, int rb, int c0, int i) {
    const int q = i >> 2, pp = i & 3;
    const int chunk = (c0 >> 3) + (pp >> 1);
    const int r0 = ra + q, r1 = rb + q;
    const int s0 = VSWZ ? swz_trv(r0) : swz_row(r0);
    const int s1 = VSWZ ? swz_trv(r1) : swz_row(r1);
    const bf16_t* a0 = img + r0 * HD + ((chunk ^ s0) << 3) + 4 * (pp & 1);
    const bf16_t* a1 = img + r1 * HD + ((chunk ^ s1) << 3) + 4 * (pp & 1);
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)a0);
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)a1);
    short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

// the same from precomputed addresses (rows ra + q / rb + q of the lane): used where the swizzle does not depend on the loop
__device__ __forceinline__ bf16x8_t lds_tr_frag_at(const bf16_t* a0, const bf16_t* a1) {
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)a0);
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)a1);
    short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ bf16x8_t pack8(const float4_t& a, const float4_t& b) {
    short8_t v;
    v[0] = (short)f32_to_bf16(a[0]); v[1] = (short)f32_to_bf16(a[1]); v[2] = (short)f32_to_bf16(a[2]); v[3] = (short)f32_to_bf16(a[3]);
    v[4] = (short)f32_to_bf16(b[0]); v[5] = (short)f32_to_bf16(b[1]); v[6] = (short)f32_to_bf16(b[2]); v[7] = (short)f32_to_bf16(b[3]);
    return __builtin_bit_cast(bf16x8_t, v);
}

// stage rows [0, nrows_pad) x 64 of one head slice into an LDS image (zero rows >= n_valid)
template <bool VSWZ, int NTHREADS = 256>
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ src, int64_t row_stride, int n_valid, int nrows_pad,
                                           bf16_t* img, int tid) {
    for (int id = tid; id < nrows_pad * 8; id += NTHREADS) {
        const int r = id >> 3, c = id & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < n_valid) v = *reinterpret_cast<const uint4*>(src + (int64_t)r * row_stride + c * 8);
        const int s = VSWZ ? swz_trv(r) : swz_row(r);
        *reinterpret_cast<uint4*>(img + r * HD + ((c ^ s) << 3)) = v;
    }
}

// ------------------------------------------------------------------------------------ forward
template <int NTP, bool DROP>  // pairs of 16-key tiles; padded key count = 32 * NTP
__global__ void __launch_bounds__(256, 2) attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse_out,
                                                       int N, int H, float scale_log2, float drop_scale, uint32_t drop_thr,
                                                       uint32_t drop_key) {
    constexpr int NKP = 32 * NTP, NT = 2 * NTP;
    __shared__ __attribute__((aligned(16))) bf16_t Ks[NKP * HD];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[NKP * HD];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int Dm = H * HD;
    const int64_t D3 = 3 * (int64_t)Dm;
    const bf16_t* base = qkv + (int64_t)b * N * D3 + h * HD;

    stage_rows<false>(base + Dm, D3, N, NKP, Ks, tid);
    stage_rows<true>(base + 2 * Dm, D3, N, NKP, Vs, tid);
    __syncthreads();

    const int nqt = (N + 15) >> 4;
    for (int qt = wave; qt < nqt; qt += 4) {
        const int q0 = qt * 16;
        const int qrow = min(q0 + i, N - 1);  // clamp: pad queries recompute a valid row, never stored
        const bf16_t* qp = base + (int64_t)qrow * D3;
        const bf16x8_t qf0 = *reinterpret_cast<const bf16x8_t*>(qp + g * 8);
        const bf16x8_t qf1 = *reinterpret_cast<const bf16x8_t*>(qp + 32 + g * 8);

        float4_t s[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            s[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ks, 16 * t + i, g), qf0, s[t], 0, 0, 0);
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ks, 16 * t + i, 4 + g), qf1, s[t], 0, 0, 0);
        }
        // lane (g,i): s[t][r] = score(query q0+i, key 16t + 4g + r).  Only the tiles that straddle or exceed N need the
        // key < N mask (wave-uniform test per tile); max is taken on raw scores (scale > 0) and scale / max are folded
        // into one FMA in front of v_exp_f32.
        const int full_tiles = N >> 4;   // tiles t < full_tiles hold only valid keys
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t >= full_tiles) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] = (16 * t + 4 * g + r < N) ? s[t][r] : -INFINITY;
            }
            mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxs = mx * scale_log2;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], scale_log2, -mxs));
                sum += s[t][r];
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        const int q = q0 + i;
        if (g == 0 && q < N) lse_out[(int64_t)bh * N + q] = (mxs + log2f(sum)) * 0.69314718055994530942f;
        const uint32_t ebase = ((uint32_t)bh * (uint32_t)N + (uint32_t)min(q, N - 1)) * (uint32_t)N;   // B*H*N*N < 2^32 (checked on the host)
        // dropout on the (still unnormalised) probabilities; 1/sum and 1/(1-rate) are applied to O (16 values) instead
        if (DROP) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint32_t e0 = ebase + (uint32_t)(16 * t + 4 * g);
                const uint32_t c0 = e0 >> 1, odd = e0 & 1u;
                const uint32_t h0 = chb_hash32(c0 ^ drop_key), h1 = chb_hash32((c0 + 1u) ^ drop_key), h2 = chb_hash32((c0 + 2u) ^ drop_key);
                // element e0 + r uses 16-bit half ((odd + r) & 1) of hash (odd + r) >> 1
                const uint32_t ue[4] = {h0 & 0xffffu, h0 >> 16, h1 & 0xffffu, h1 >> 16};
                const uint32_t uo[4] = {h0 >> 16, h1 & 0xffffu, h1 >> 16, h2 & 0xffffu};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t u = odd ? uo[r] : ue[r];
                    s[t][r] = (u >= drop_thr) ? s[t][r] : 0.f;
                }
            }
        }
        const float oscale = DROP ? inv * drop_scale : inv;
        // O^T[d][q] = sum_key V^T[d][key] P^T[key][q]; k-slot (g, j): j<4 -> key 32u+4g+j, j>=4 -> key 32u+16+4g+(j-4)
        float4_t oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NTP; ++u) {
            const bf16x8_t pf = pack8(s[2 * u], s[2 * u + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8_t vf = lds_tr_frag<true>(Vs, 32 * u + 4 * g, 32 * u + 16 + 4 * g, 16 * dt, i);
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt], 0, 0, 0);
            }
        }
        if (q < N) {
            bf16_t* op = o + ((int64_t)b * N + q) * Dm + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 w;
                w.x = pack_bf16x2(oacc[dt][0] * oscale, oacc[dt][1] * oscale);
                w.y = pack_bf16x2(oacc[dt][2] * oscale, oacc[dt][3] * oscale);
                *reinterpret_cast<uint2*>(op + 16 * dt + 4 * g) = w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------ backward
// LDS: K, V, Q, dO images [NP][64] (row-read swizzle; transposed reads take a 2-way conflict),
// dS double buffer [2][32][NP + 8], lse*log2e and delta per query.
template <int NTP, bool DROP, int NW, bool DBIAS = true>   // NW waves per workgroup: 8, or 16 (4 per SIMD, <= 128 registers) for 129..224 tokens
__global__ void __launch_bounds__(NW * 64, NW / 4) attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o, const bf16_t* __restrict__ d_o,
                                                       const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int N, int H, float scale,
                                                       float scale_log2, float drop_scale, uint32_t drop_thr, uint32_t drop_key,
                                                       float* __restrict__ dbias) {
    constexpr int NP = 32 * NTP, NT = 2 * NTP;
    constexpr int MT = (NT + NW - 1) / NW;   // key tiles owned by one wave (wave w of NW: tiles w, w + NW, ...)
    constexpr int NTHR = NW * 64;
    constexpr int DSLD = NP + 8;       // dS row stride (elements)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* Vs = Ks + NP * HD;
    bf16_t* Qs = Vs + NP * HD;
    bf16_t* Gs = Qs + NP * HD;                      // dO
    bf16_t* dSs = Gs + NP * HD;                     // [2][32][DSLD]
    float* lse2 = reinterpret_cast<float*>(dSs + 2 * 32 * DSLD);
    float* delta = lse2 + NP;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int Dm = H * HD;
    const int64_t D3 = 3 * (int64_t)Dm;
    const bf16_t* base = qkv + (int64_t)b * N * D3 + h * HD;
    const bf16_t* obase = o + (int64_t)b * N * Dm + h * HD;
    const bf16_t* gbase = d_o + (int64_t)b * N * Dm + h * HD;

    stage_rows<false, NTHR>(base, D3, N, NP, Qs, tid);
    stage_rows<false, NTHR>(base + Dm, D3, N, NP, Ks, tid);
    stage_rows<false, NTHR>(base + 2 * Dm, D3, N, NP, Vs, tid);
    // dO image + delta[q] = sum_d dO*O (8 threads per row, one 16-byte chunk each)
    for (int id = tid; id < ((NP * 8 + NTHR - 1) & ~(NTHR - 1)); id += NTHR) {   // full waves only: the reduction below shuffles
        const int r = id >> 3, c = id & 7;
        uint4 gv = make_uint4(0, 0, 0, 0), ov = make_uint4(0, 0, 0, 0);
        if (r < N && r < NP) {
            gv = *reinterpret_cast<const uint4*>(gbase + (int64_t)r * Dm + c * 8);
            ov = *reinterpret_cast<const uint4*>(obase + (int64_t)r * Dm + c * 8);
        }
        if (r < NP) *reinterpret_cast<uint4*>(Gs + r * HD + ((c ^ swz_row(r)) << 3)) = gv;
        const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d += bf16_to_f32((bf16_t)(gw[k] & 0xffff)) * bf16_to_f32((bf16_t)(ow[k] & 0xffff));
            d += bf16_to_f32((bf16_t)(gw[k] >> 16)) * bf16_to_f32((bf16_t)(ow[k] >> 16));
        }
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        if (c == 0 && r < NP) delta[r] = d;
    }
    for (int r = tid; r < NP; r += NTHR) lse2[r] = r < N ? lse[(int64_t)bh * N + r] * 1.44269504088896340736f : INFINITY;
    __syncthreads();

    float4_t dk[4][MT], dv[4][MT];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int c = 0; c < MT; ++c) {
            dk[dt][c] = (float4_t){0.f, 0.f, 0.f, 0.f};
            dv[dt][c] = (float4_t){0.f, 0.f, 0.f, 0.f};
        }

    float4_t dqsum = (float4_t){0.f, 0.f, 0.f, 0.f};   // column sums of this wave's dQ tiles (bias gradient of the query projection)
    // this wave's K / V row fragments (B operands of S and dP) never change across query blocks: keep them in registers
    bf16x8_t kfr[MT][2], vfr[MT][2];
#pragma unroll
    for (int c = 0; c < MT; ++c)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int t = min(wave + NW * c, NT - 1);
            kfr[c][ks] = lds_row_frag(Ks, 16 * t + i, 4 * ks + g);
            vfr[c][ks] = lds_row_frag(Vs, 16 * t + i, 4 * ks + g);
        }
    // Lane-constant LDS element offsets.  Query blocks start at multiples of 32 rows and key pairs at multiples of 32, so the row
    // swizzle ((r >> 1) & 7) of every fragment row depends on the lane only: all address arithmetic leaves the block loop.
    const int lq = i >> 2, lpp = i & 3;
    int rowoff[2];            // row fragments of Q / dO: row (block row 16 qs) + i, chunk 4 ks + g
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) rowoff[ks] = i * HD + (((4 * ks + g) ^ swz_row(i)) << 3);
    int troff[4];             // transposed fragments of Q / dO: rows 4g + lq (and + 16), columns 16 dt + 4 lpp ..
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) troff[dt] = (4 * g + lq) * HD + (((2 * dt + (lpp >> 1)) ^ swz_row(4 * g + lq)) << 3) + 4 * (lpp & 1);
    // round `it` runs phase A of query block `it` (scores, dS, dK/dV) and phase B of block `it - 1` (dQ from the dS tile
    // published one round earlier) between the same pair of barriers: one barrier per block instead of two
    for (int it = 0; it <= NTP; ++it) {
        const int q0 = 32 * it;
        bf16_t* dSb = dSs + (it & 1) * 32 * DSLD;
        if (it < NTP) {
        // ---- phase A: this wave's key tiles against the 32 queries of the block
        bf16x8_t qa[2][2], ga[2][2];  // [query sub-tile][k-step]: A operands, rows = queries
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qa[qs][ks] = *reinterpret_cast<const bf16x8_t*>(Qs + (q0 + 16 * qs) * HD + rowoff[ks]);
                ga[qs][ks] = *reinterpret_cast<const bf16x8_t*>(Gs + (q0 + 16 * qs) * HD + rowoff[ks]);
            }
        float l2[2][4], dl[2][4];
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                l2[qs][r] = lse2[q0 + 16 * qs + 4 * g + r];
                dl[qs][r] = delta[q0 + 16 * qs + 4 * g + r];
            }
        // dropout element index ((b*H+h)*N + q)*N + key: one multiply per block, rows by adding multiples of N.  Queries and keys
        // past N need no clamp: their probabilities are exactly zero (lse = +inf / key mask), whatever the mask bit says.
        uint32_t erow[2][4];
        {
            const uint32_t e0 = ((uint32_t)bh * (uint32_t)N + (uint32_t)(q0 + 4 * g)) * (uint32_t)N;
            const uint32_t un = (uint32_t)__builtin_amdgcn_readfirstlane(N);
            erow[0][0] = e0;
            erow[0][1] = erow[0][0] + un;
            erow[0][2] = erow[0][1] + un;
            erow[0][3] = erow[0][2] + un;
            const uint32_t un16 = un << 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) erow[1][r] = erow[0][r] + un16;
        }
#pragma unroll
        for (int c = 0; c < MT; ++c) {
            const int t = wave + NW * c;
            if (t < NT) {  // wave-uniform
                const int key = 16 * t + i;
                const bool tile_full = 16 * t + 16 <= N;   // wave-uniform
                float4_t pd[2], ds[2];
#pragma unroll
                for (int qs = 0; qs < 2; ++qs) {
                    float4_t sv = (float4_t){0.f, 0.f, 0.f, 0.f}, dp = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        // D[row = query 4g+r][col = key i]
                        sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[qs][ks], kfr[c][ks], sv, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[qs][ks], vfr[c][ks], dp, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], scale_log2, -l2[qs][r]));
                        if (!tile_full) p = (key < N) ? p : 0.f;
                        float keepc = 1.0f;
                        if (DROP) {
                            // element ((b*H+h)*N + q)*N + key as 32-bit arithmetic; one row base per (qs, r), one add per key
                            const uint32_t e = erow[qs][r] + (uint32_t)key;
                            const uint32_t hsh = chb_hash32((e >> 1) ^ drop_key);
                            const uint32_t u = (e & 1u) ? (hsh >> 16) : (hsh & 0xffffu);
                            keepc = (u >= drop_thr) ? drop_scale : 0.f;
                        }
                        pd[qs][r] = p * keepc;                                   // dropped probabilities (for dV)
                        ds[qs][r] = p * (dp[r] * keepc - dl[qs][r]) * scale;     // d(scores) incl. 1/sqrt(hd)
                    }
                }
                // contraction over the 32 queries: k-slot (g, j): j<4 -> q0+4g+j, j>=4 -> q0+16+4g+(j-4)
                const bf16x8_t pf = pack8(pd[0], pd[1]);
                const bf16x8_t sf = pack8(ds[0], ds[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8_t gt = lds_tr_frag_at(Gs + q0 * HD + troff[dt], Gs + (q0 + 16) * HD + troff[dt]);
                    const bf16x8_t qt = lds_tr_frag_at(Qs + q0 * HD + troff[dt], Qs + (q0 + 16) * HD + troff[dt]);
                    dv[dt][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt, pf, dv[dt][c], 0, 0, 0);  // dV^T[d][key]
                    dk[dt][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, sf, dk[dt][c], 0, 0, 0);  // dK^T[d][key]
                }
                // dS tile -> LDS [query][key] for the dQ product
#pragma unroll
                for (int qs = 0; qs < 2; ++qs)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dSb[(16 * qs + 4 * g + r) * DSLD + key] = f32_to_bf16(ds[qs][r]);
            }
        }
        }   // phase A
        // ---- phase B (block it - 1): dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]; wave w owns query sub-tile w>>2, d-tile w&3
        if (it > 0 && wave < 8) {   // wave-uniform
            const int q0 = 32 * (it - 1);
            const bf16_t* dSb = dSs + ((it - 1) & 1) * 32 * DSLD;
            const int qs = wave >> 2, dtw = wave & 3;
            float4_t dq = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NTP; ++u) {
                const bf16x8_t kt = lds_tr_frag<false>(Ks, 32 * u + 8 * g, 32 * u + 8 * g + 4, 16 * dtw, i);  // A[row d][k = key 32u+8g+j]
                const bf16x8_t sb = *reinterpret_cast<const bf16x8_t*>(dSb + (16 * qs + i) * DSLD + 32 * u + 8 * g);  // B[k][col q]
                dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, sb, dq, 0, 0, 0);
            }
            const int q = q0 + 16 * qs + i;
            if (q < N) {
                uint2 w;
                w.x = pack_bf16x2(dq[0], dq[1]);
                w.y = pack_bf16x2(dq[2], dq[3]);
                *reinterpret_cast<uint2*>(dqkv + ((int64_t)b * N + q) * D3 + h * HD + 16 * dtw + 4 * g) = w;
                if (DBIAS) dqsum += dq;
            }
        }
        __syncthreads();   // dS of block `it` is published; dS buffer (it - 1) & 1 is free for block it + 1
    }
    // dK^T / dV^T accumulators: lane (g,i) reg r = [d = 16dt + 4g + r][key = 16t + i]
#pragma unroll
    for (int c = 0; c < MT; ++c) {
        const int t = wave + NW * c;
        const int key = 16 * t + i;
        if (t < NT && key < N) {
            bf16_t* kp = dqkv + ((int64_t)b * N + key) * D3 + Dm + h * HD;
            bf16_t* vp = kp + Dm;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 w;
                w.x = pack_bf16x2(dk[dt][c][0], dk[dt][c][1]);
                w.y = pack_bf16x2(dk[dt][c][2], dk[dt][c][3]);
                *reinterpret_cast<uint2*>(kp + 16 * dt + 4 * g) = w;
                w.x = pack_bf16x2(dv[dt][c][0], dv[dt][c][1]);
                w.y = pack_bf16x2(dv[dt][c][2], dv[dt][c][3]);
                *reinterpret_cast<uint2*>(vp + 16 * dt + 4 * g) = w;
            }
        }
    }
    if (DBIAS && dbias) {
        // bias gradients of the fused QKV projection = column sums of dqkv over this head's rows.  Every wave folds its tiles over
        // the 16 lanes that hold different queries / keys and leaves 192 column partials in LDS (the operand images are dead after the
        // loop's last barrier); 192 threads add the waves' partials and write the head's sums ONCE: to this batch element's row of
        // the workspace (dbias_rows: plain stores, reduced over the batch by dbias_reduce_kernel) or with one atomic per column.
        float* red = reinterpret_cast<float*>(smem_raw);       // [NW][192]: q | k | v columns of this head (over the dead K image)
        if (wave < 8) {                                        // dQ tiles belong to waves 0..7: d-tile wave & 3, query sub-tile wave >> 2
            const int dtw = wave & 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = dqsum[r];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                if (i == 0) red[wave * 192 + 16 * dtw + 4 * g + r] = v;
            }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float vk = 0.f, vv = 0.f;
#pragma unroll
                for (int c = 0; c < MT; ++c) {
                    const int key = 16 * (wave + NW * c) + i;
                    if (wave + NW * c < NT && key < N) {
                        vk += dk[dt][c][r];
                        vv += dv[dt][c][r];
                    }
                }
                vk += __shfl_xor(vk, 1, 64); vk += __shfl_xor(vk, 2, 64); vk += __shfl_xor(vk, 4, 64); vk += __shfl_xor(vk, 8, 64);
                vv += __shfl_xor(vv, 1, 64); vv += __shfl_xor(vv, 2, 64); vv += __shfl_xor(vv, 4, 64); vv += __shfl_xor(vv, 8, 64);
                if (i == 0) {
                    red[wave * 192 + 64 + 16 * dt + 4 * g + r] = vk;
                    red[wave * 192 + 128 + 16 * dt + 4 * g + r] = vv;
                }
            }
        __syncthreads();
        if (wave < 3) {                                        // 192 threads: part = wave, column = lane
            const int part = wave, col = lane;
            float sum = 0.f;
            if (part == 0) {
                const int dtw = col >> 4;                      // the two waves that own this d-tile (query sub-tiles 0 and 1)
                sum = red[dtw * 192 + col] + red[(dtw + 4) * 192 + col];
            } else {
#pragma unroll
                for (int w = 0; w < NW; ++w) sum += red[w * 192 + part * 64 + col];
            }
            dbias[(int64_t)b * 3 * Dm + part * Dm + h * HD + col] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------ backward, long sequences
// N > 224 (ViT at 384^2: N = 577): Q, K, V, dO of a head no longer fit LDS together.  Two passes, no atomics, fp32
// accumulation throughout (FlashAttention-2 style split):
//   pass 1 (dK, dV): workgroup = (head, 128 keys); a wave keeps its 16 keys' K/V fragments and dK^T/dV^T in registers
//                    and walks the query blocks, Q / dO / lse / delta streamed through a double-buffered LDS block.
//   pass 2 (dQ)    : workgroup = (head, 128 queries); a wave keeps its 16 queries' Q / dO fragments in registers and
//                    walks the key chunks (K / V double-buffered in LDS); scores are recomputed transposed
//                    (S^T = K.Q^T) so dS^T feeds the dQ MFMA straight from the accumulators.

// dropout keep factors for 4 consecutive elements e0 .. e0+3 (three hashes cover the 16-bit halves of both parities)
__device__ __forceinline__ void keep4(uint32_t e0, uint32_t drop_key, uint32_t drop_thr, float drop_scale, float out[4]) {
    const uint32_t c0 = e0 >> 1, odd = e0 & 1u;
    const uint32_t h0 = chb_hash32(c0 ^ drop_key), h1 = chb_hash32((c0 + 1u) ^ drop_key), h2 = chb_hash32((c0 + 2u) ^ drop_key);
    const uint32_t ue[4] = {h0 & 0xffffu, h0 >> 16, h1 & 0xffffu, h1 >> 16};
    const uint32_t uo[4] = {h0 >> 16, h1 & 0xffffu, h1 >> 16, h2 & 0xffffu};
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = ((odd ? uo[r] : ue[r]) >= drop_thr) ? drop_scale : 0.f;
}

__device__ __forceinline__ float dot8_bf16(const uint4& a, const uint4& b) {
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    float d = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        d += bf16_to_f32((bf16_t)(aw[k] & 0xffff)) * bf16_to_f32((bf16_t)(bw[k] & 0xffff));
        d += bf16_to_f32((bf16_t)(aw[k] >> 16)) * bf16_to_f32((bf16_t)(bw[k] >> 16));
    }
    return d;
}

// ------------------------------------------------------------------------------------ forward, long sequences
// N > 224: the score row of a query no longer fits the register file next to K/V in LDS.  Workgroup = (head, 128
// queries), a wave owns 16 queries; K / V stream through a double-buffered LDS chunk of 64 keys and the softmax is
// computed online (running max / sum per query, accumulator rescaled when the max moves).  Same dropout element
// index and the same operand rounding points as the resident kernel (un-normalised P rounded to bf16, 1/sum and
// 1/(1-rate) applied to the fp32 output), so the two agree to bf16 rounding of P.
template <bool DROP>
__global__ void __launch_bounds__(512, 2) attn_fwd_stream_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                              float* __restrict__ lse_out, int N, int H, float scale_log2,
                                                              float drop_scale, uint32_t drop_thr, uint32_t drop_key) {
    constexpr int KC = 64;
    __shared__ __attribute__((aligned(16))) bf16_t Kb[2][KC * HD];
    __shared__ __attribute__((aligned(16))) bf16_t Vb[2][KC * HD];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int Dm = H * HD;
    const int64_t D3 = 3 * (int64_t)Dm;
    const bf16_t* base = qkv + (int64_t)b * N * D3 + h * HD;

    const int q0w = 128 * blockIdx.y + 16 * wave;
    const int q = q0w + i, qc = min(q, N - 1);
    const bool wave_live = q0w < N;   // wave-uniform
    const bf16_t* qp = base + (int64_t)qc * D3;
    const bf16x8_t qf0 = *reinterpret_cast<const bf16x8_t*>(qp + g * 8);
    const bf16x8_t qf1 = *reinterpret_cast<const bf16x8_t*>(qp + 32 + g * 8);
    const uint32_t ebase = ((uint32_t)bh * (uint32_t)N + (uint32_t)qc) * (uint32_t)N;

    float m_run = -INFINITY;   // running max of the raw scores of this query (identical in the 4 lanes g of a query)
    float l_run = 0.f;         // this lane's share of the running sum (keys 4g+r of every tile)
    float4_t oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int sr = tid >> 3, sc = tid & 7;   // staging: 64 rows x 8 chunks, K and V
    uint4 skv = make_uint4(0, 0, 0, 0), svv = make_uint4(0, 0, 0, 0);
    auto stage_load = [&](int k0) {
        const int r = k0 + sr;
        skv = make_uint4(0, 0, 0, 0);
        svv = make_uint4(0, 0, 0, 0);
        if (r < N) {
            const bf16_t* kp = base + (int64_t)r * D3 + Dm + sc * 8;
            skv = *reinterpret_cast<const uint4*>(kp);
            svv = *reinterpret_cast<const uint4*>(kp + Dm);
        }
    };
    auto stage_store = [&](int buf) {
        *reinterpret_cast<uint4*>(Kb[buf] + sr * HD + ((sc ^ swz_row(sr)) << 3)) = skv;
        *reinterpret_cast<uint4*>(Vb[buf] + sr * HD + ((sc ^ swz_trv(sr)) << 3)) = svv;
    };

    const int nkc = (N + KC - 1) / KC;
    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int kc = 0; kc < nkc; ++kc) {
        const int cur = kc & 1, k0 = KC * kc;
        if (kc + 1 < nkc) stage_load(k0 + KC);
        if (wave_live) {
            const bf16_t* Ks = Kb[cur];
            const bf16_t* Vs = Vb[cur];
            // S^T tiles of the chunk: lane (g,i) reg r = score(query i, key k0 + 16t + 4g + r)
            float4_t s[KC / 16];
            float cmax = -INFINITY;
#pragma unroll
            for (int t = 0; t < KC / 16; ++t) {
                s[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
                if (k0 + 16 * t < N) {   // wave-uniform
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ks, 16 * t + i, g), qf0, s[t], 0, 0, 0);
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ks, 16 * t + i, 4 + g), qf1, s[t], 0, 0, 0);
                    if (k0 + 16 * t + 16 > N) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[t][r] = (k0 + 16 * t + 4 * g + r < N) ? s[t][r] : -INFINITY;
                    }
                } else {
                    s[t] = (float4_t){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
                cmax = fmaxf(cmax, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
            }
            cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
            const float m_new = fmaxf(m_run, cmax);           // finite: every chunk holds at least one valid key
            const float corr = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2);   // first chunk: exp2(-inf) = 0
            m_run = m_new;
            const float mxs = m_new * scale_log2;
            l_run *= corr;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt] *= corr;
#pragma unroll
            for (int t = 0; t < KC / 16; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], scale_log2, -mxs));
                    l_run += s[t][r];
                }
                if (DROP) {
                    float keepc[4];
                    keep4(ebase + (uint32_t)(k0 + 16 * t + 4 * g), drop_key, drop_thr, 1.0f, keepc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[t][r] *= keepc[r];
                }
            }
            // O^T[d][q] += V^T[d][key] P^T[key][q]; k-slot (g, j): j<4 -> key 32u+4g+j, j>=4 -> key 32u+16+4g+(j-4)
#pragma unroll
            for (int u = 0; u < KC / 32; ++u) {
                if (k0 + 32 * u < N) {   // wave-uniform
                    const bf16x8_t pf = pack8(s[2 * u], s[2 * u + 1]);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const bf16x8_t vf = lds_tr_frag<true>(Vs, 32 * u + 4 * g, 32 * u + 16 + 4 * g, 16 * dt, i);
                        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt], 0, 0, 0);
                    }
                }
            }
        }
        if (kc + 1 < nkc) stage_store(cur ^ 1);
        __syncthreads();
    }
    if (wave_live) {
        float sum = l_run;
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float oscale = (DROP ? drop_scale : 1.0f) / sum;
        if (q < N) {
            if (g == 0) lse_out[(int64_t)bh * N + q] = (m_run * scale_log2 + log2f(sum)) * 0.69314718055994530942f;
            bf16_t* op = o + ((int64_t)b * N + q) * Dm + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 w;
                w.x = pack_bf16x2(oacc[dt][0] * oscale, oacc[dt][1] * oscale);
                w.y = pack_bf16x2(oacc[dt][2] * oscale, oacc[dt][3] * oscale);
                *reinterpret_cast<uint2*>(op + 16 * dt + 4 * g) = w;
            }
        }
    }
}

template <bool DROP>
__global__ void __launch_bounds__(512, 2) attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                           const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dqkv, int N, int H, float scale, float scale_log2,
                                                           float drop_scale, uint32_t drop_thr, uint32_t drop_key,
                                                           float* __restrict__ dbias) {
    __shared__ __attribute__((aligned(16))) bf16_t Qb[2][32 * HD];
    __shared__ __attribute__((aligned(16))) bf16_t Gb[2][32 * HD];
    __shared__ float l2b[2][32], dlb[2][32];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int Dm = H * HD;
    const int64_t D3 = 3 * (int64_t)Dm;
    const bf16_t* base = qkv + (int64_t)b * N * D3 + h * HD;
    const bf16_t* obase = o + (int64_t)b * N * Dm + h * HD;
    const bf16_t* gbase = d_o + (int64_t)b * N * Dm + h * HD;

    const int kt0 = 128 * blockIdx.y + 16 * wave;    // first key of this wave's tile
    const int key = kt0 + i;
    const bool tile_live = kt0 < N, tile_full = kt0 + 16 <= N;   // wave-uniform
    bf16x8_t kfr[2], vfr[2];
    {
        const bf16_t* kp = base + (int64_t)min(key, N - 1) * D3 + Dm;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kfr[ks] = *reinterpret_cast<const bf16x8_t*>(kp + 32 * ks + 8 * g);
            vfr[ks] = *reinterpret_cast<const bf16x8_t*>(kp + Dm + 32 * ks + 8 * g);
        }
    }
    float4_t dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        dk[dt] = (float4_t){0.f, 0.f, 0.f, 0.f};
        dv[dt] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }

    // staging roles: threads 0..255 bring dO (+ O for delta), threads 256..511 bring Q (+ lse); 8 threads per row
    const int sr = (tid & 255) >> 3, sc = tid & 7;
    const bool role_g = tid < 256;   // wave-uniform
    uint4 sv0 = make_uint4(0, 0, 0, 0), sv1 = make_uint4(0, 0, 0, 0);
    float sl = 0.f;
    auto stage_load = [&](int q0) {
        const int r = q0 + sr;
        sv0 = make_uint4(0, 0, 0, 0);
        sv1 = make_uint4(0, 0, 0, 0);
        sl = INFINITY;
        if (r < N) {
            if (role_g) {
                sv0 = *reinterpret_cast<const uint4*>(gbase + (int64_t)r * Dm + sc * 8);
                sv1 = *reinterpret_cast<const uint4*>(obase + (int64_t)r * Dm + sc * 8);
            } else {
                sv0 = *reinterpret_cast<const uint4*>(base + (int64_t)r * D3 + sc * 8);
                if (sc == 0) sl = lse[(int64_t)bh * N + r] * 1.44269504088896340736f;
            }
        }
    };
    auto stage_store = [&](int buf) {
        bf16_t* dst = (role_g ? Gb[buf] : Qb[buf]) + sr * HD + ((sc ^ swz_row(sr)) << 3);
        *reinterpret_cast<uint4*>(dst) = sv0;
        if (role_g) {
            float d = dot8_bf16(sv0, sv1);
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            if (sc == 0) dlb[buf][sr] = d;
        } else if (sc == 0) {
            l2b[buf][sr] = sl;
        }
    };

    const int nqb = (N + 31) >> 5;
    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int it = 0; it < nqb; ++it) {
        const int cur = it & 1, q0 = 32 * it;
        if (it + 1 < nqb) stage_load(q0 + 32);
        if (tile_live) {
            const bf16_t* Qs = Qb[cur];
            const bf16_t* Gs = Gb[cur];
            float4_t pd[2], ds[2];
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) {
                float4_t sv = (float4_t){0.f, 0.f, 0.f, 0.f}, dp = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    // D[row = query 4g+r][col = key i]
                    sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Qs, 16 * qs + i, 4 * ks + g), kfr[ks], sv, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Gs, 16 * qs + i, 4 * ks + g), vfr[ks], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = 16 * qs + 4 * g + r;
                    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], scale_log2, -l2b[cur][ql]));   // pad queries: lse = +inf -> 0
                    if (!tile_full) p = (key < N) ? p : 0.f;
                    float keepc = 1.0f;
                    if (DROP) {
                        const uint32_t e = ((uint32_t)bh * (uint32_t)N + (uint32_t)min(q0 + ql, N - 1)) * (uint32_t)N + (uint32_t)min(key, N - 1);
                        const uint32_t hsh = chb_hash32((e >> 1) ^ drop_key);
                        const uint32_t u = (e & 1u) ? (hsh >> 16) : (hsh & 0xffffu);
                        keepc = (u >= drop_thr) ? drop_scale : 0.f;
                    }
                    pd[qs][r] = p * keepc;
                    ds[qs][r] = p * (dp[r] * keepc - dlb[cur][ql]) * scale;
                }
            }
            // contraction over the 32 queries: k-slot (g, j): j<4 -> query 4g+j, j>=4 -> query 16+4g+(j-4)
            const bf16x8_t pf = pack8(pd[0], pd[1]);
            const bf16x8_t sf = pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8_t gt = lds_tr_frag<false>(Gs, 4 * g, 16 + 4 * g, 16 * dt, i);
                const bf16x8_t qt = lds_tr_frag<false>(Qs, 4 * g, 16 + 4 * g, 16 * dt, i);
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt, pf, dv[dt], 0, 0, 0);  // dV^T[d][key]
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, sf, dk[dt], 0, 0, 0);  // dK^T[d][key]
            }
        }
        if (it + 1 < nqb) stage_store(cur ^ 1);   // last read in round it - 1, all waves are past that barrier
        __syncthreads();
    }
    // lane (g,i) reg r = [d = 16dt + 4g + r][key]
    if (tile_live && key < N) {
        bf16_t* kp = dqkv + ((int64_t)b * N + key) * D3 + Dm + h * HD;
        bf16_t* vp = kp + Dm;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            uint2 w;
            w.x = pack_bf16x2(dk[dt][0], dk[dt][1]);
            w.y = pack_bf16x2(dk[dt][2], dk[dt][3]);
            *reinterpret_cast<uint2*>(kp + 16 * dt + 4 * g) = w;
            w.x = pack_bf16x2(dv[dt][0], dv[dt][1]);
            w.y = pack_bf16x2(dv[dt][2], dv[dt][3]);
            *reinterpret_cast<uint2*>(vp + 16 * dt + 4 * g) = w;
        }
    }
    if (dbias && tile_live) {
        const float live = key < N ? 1.f : 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float vk = dk[dt][r] * live, vv = dv[dt][r] * live;
                vk += __shfl_xor(vk, 1, 64); vk += __shfl_xor(vk, 2, 64); vk += __shfl_xor(vk, 4, 64); vk += __shfl_xor(vk, 8, 64);
                vv += __shfl_xor(vv, 1, 64); vv += __shfl_xor(vv, 2, 64); vv += __shfl_xor(vv, 4, 64); vv += __shfl_xor(vv, 8, 64);
                if (i == 0) {
                    atomicAdd(dbias + Dm + h * HD + 16 * dt + 4 * g + r, vk);
                    atomicAdd(dbias + 2 * Dm + h * HD + 16 * dt + 4 * g + r, vv);
                }
            }
    }
}

template <bool DROP>
__global__ void __launch_bounds__(512, 2) attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                          const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                          bf16_t* __restrict__ dqkv, int N, int H, float scale, float scale_log2,
                                                          float drop_scale, uint32_t drop_thr, uint32_t drop_key,
                                                          float* __restrict__ dbias) {
    constexpr int KC = 64;   // keys per staged chunk
    __shared__ __attribute__((aligned(16))) bf16_t Kb[2][KC * HD];
    __shared__ __attribute__((aligned(16))) bf16_t Vb[2][KC * HD];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int Dm = H * HD;
    const int64_t D3 = 3 * (int64_t)Dm;
    const bf16_t* base = qkv + (int64_t)b * N * D3 + h * HD;

    const int q0w = 128 * blockIdx.y + 16 * wave;
    const int q = q0w + i, qc = min(q, N - 1);
    const bool wave_live = q0w < N;   // wave-uniform
    // B operands [k = d][col = query i] of S^T = K.Q^T and dP^T = V.dO^T; delta = sum_d dO*O
    bf16x8_t qb[2], gb[2];
    float dl = 0.f;
    {
        const bf16_t* qp = base + (int64_t)qc * D3;
        const bf16_t* gp = d_o + ((int64_t)b * N + qc) * Dm + h * HD;
        const bf16_t* op = o + ((int64_t)b * N + qc) * Dm + h * HD;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qb[ks] = *reinterpret_cast<const bf16x8_t*>(qp + 32 * ks + 8 * g);
            const uint4 gv = *reinterpret_cast<const uint4*>(gp + 32 * ks + 8 * g);
            const uint4 ov = *reinterpret_cast<const uint4*>(op + 32 * ks + 8 * g);
            gb[ks] = __builtin_bit_cast(bf16x8_t, gv);
            dl += dot8_bf16(gv, ov);
        }
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
    }
    const float l2 = lse[(int64_t)bh * N + qc] * 1.44269504088896340736f;
    const uint32_t ebase = ((uint32_t)bh * (uint32_t)N + (uint32_t)qc) * (uint32_t)N;

    float4_t dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int sr = tid >> 3, sc = tid & 7;   // staging: 64 rows x 8 chunks, K and V
    uint4 skv = make_uint4(0, 0, 0, 0), svv = make_uint4(0, 0, 0, 0);
    auto stage_load = [&](int k0) {
        const int r = k0 + sr;
        skv = make_uint4(0, 0, 0, 0);
        svv = make_uint4(0, 0, 0, 0);
        if (r < N) {
            const bf16_t* kp = base + (int64_t)r * D3 + Dm + sc * 8;
            skv = *reinterpret_cast<const uint4*>(kp);
            svv = *reinterpret_cast<const uint4*>(kp + Dm);
        }
    };
    auto stage_store = [&](int buf) {
        const int off = sr * HD + ((sc ^ swz_row(sr)) << 3);
        *reinterpret_cast<uint4*>(Kb[buf] + off) = skv;
        *reinterpret_cast<uint4*>(Vb[buf] + off) = svv;
    };

    const int nkc = (N + KC - 1) / KC;
    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int kc = 0; kc < nkc; ++kc) {
        const int cur = kc & 1, k0 = KC * kc;
        if (kc + 1 < nkc) stage_load(k0 + KC);
        if (wave_live) {
            const bf16_t* Ks = Kb[cur];
            const bf16_t* Vs = Vb[cur];
#pragma unroll
            for (int u = 0; u < KC / 32; ++u) {
                if (k0 + 32 * u < N) {   // wave-uniform
                    float4_t ds[2];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int t = 2 * u + hf;
                        const int key0 = k0 + 16 * t + 4 * g;
                        float4_t sv = (float4_t){0.f, 0.f, 0.f, 0.f}, dp = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            // D[row = key 4g+r][col = query i]
                            sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ks, 16 * t + i, 4 * ks + g), qb[ks], sv, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Vs, 16 * t + i, 4 * ks + g), gb[ks], dp, 0, 0, 0);
                        }
                        float keepc[4] = {1.f, 1.f, 1.f, 1.f};
                        if (DROP) keep4(ebase + (uint32_t)key0, drop_key, drop_thr, drop_scale, keepc);
                        const bool tile_full = k0 + 16 * t + 16 <= N;   // wave-uniform
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], scale_log2, -l2));
                            if (!tile_full) p = (key0 + r < N) ? p : 0.f;
                            ds[hf][r] = p * (dp[r] * keepc[r] - dl) * scale;
                        }
                    }
                    // dQ^T[d][q] += K^T[d][key] dS^T[key][q]; k-slot (g, j): j<4 -> key 32u+4g+j, j>=4 -> key 32u+16+4g+(j-4)
                    const bf16x8_t sf = pack8(ds[0], ds[1]);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const bf16x8_t kt = lds_tr_frag<false>(Ks, 32 * u + 4 * g, 32 * u + 16 + 4 * g, 16 * dt, i);
                        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, sf, dq[dt], 0, 0, 0);
                    }
                }
            }
        }
        if (kc + 1 < nkc) stage_store(cur ^ 1);
        __syncthreads();
    }
    // lane (g,i) reg r = dQ[query i][d = 16dt + 4g + r]
    if (q < N) {
        bf16_t* qp = dqkv + ((int64_t)b * N + q) * D3 + h * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            uint2 w;
            w.x = pack_bf16x2(dq[dt][0], dq[dt][1]);
            w.y = pack_bf16x2(dq[dt][2], dq[dt][3]);
            *reinterpret_cast<uint2*>(qp + 16 * dt + 4 * g) = w;
        }
    }
    if (dbias && wave_live) {
        const float live = q < N ? 1.f : 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = dq[dt][r] * live;
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                if (i == 0) atomicAdd(dbias + h * HD + 16 * dt + 4 * g + r, v);
            }
    }
}

template <int NTP>
constexpr size_t bwd_lds_bytes() {
    return (size_t)(4 * 32 * NTP * HD + 2 * 32 * (32 * NTP + 8)) * sizeof(bf16_t) + (size_t)2 * 32 * NTP * sizeof(float);
}

// dbias[c] += sum over the B rows of ws[b][c]: (C / 256) x 16 workgroups, each column finishes with 16 atomics
__global__ void __launch_bounds__(256) dbias_reduce_kernel(const float* __restrict__ ws, int B, int C, float* __restrict__ dbias) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float sum = 0.f;
    for (int b = blockIdx.y; b < B; b += gridDim.y) sum += ws[(int64_t)b * C + c];
    atomicAdd(dbias + c, sum);
}

}  // namespace

extern "C" {

int chb_attention_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int hd, float drop_rate, uint32_t drop_key,
                      void* stream) {
    if (!qkv || !o || !lse || B < 0 || N <= 0 || H <= 0 || drop_rate < 0.f || drop_rate >= 1.f) return CHB_EINVAL;
    if (hd != HD) return CHB_EUNSUPPORTED;
    if ((double)B * H * N * N >= 4294967296.0) return CHB_EUNSUPPORTED;   // dropout element index is 32-bit
    if (B == 0) return CHB_OK;
    const float scale_log2 = 1.44269504088896340736f / sqrtf((float)hd);
    const float ds = 1.0f / (1.0f - drop_rate);
    const uint32_t thr = drop_rate > 0.f ? chb_drop_threshold(drop_rate) : 0u;
    const dim3 grid(B * H), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bf16_t* in = (const bf16_t*)qkv;
    bf16_t* out = (bf16_t*)o;
    // Short sequences (N <= 128): K / V of the head resident in LDS, whole score row in registers.  Otherwise the streaming
    // kernel with online softmax (any N; at N = 197 it runs 4 waves per SIMD against 2 and measures ~10 % faster).
    // CHB_ATTN_FWD_ALGO = 1 | 2 forces resident (N <= 224) | streaming; the parity tests cross-check the two.
    const int algo = chb_option(CHB_OPT_ATTN_FWD_ALGO);
    if (N > 224 || algo == 2 || (algo != 1 && N > 128)) {
        const dim3 grid2(B * H, (N + 127) / 128);
        if (thr) hipLaunchKernelGGL((attn_fwd_stream_kernel<true>), grid2, dim3(512), 0, s, in, out, lse, N, H, scale_log2, ds, thr, drop_key);
        else hipLaunchKernelGGL((attn_fwd_stream_kernel<false>), grid2, dim3(512), 0, s, in, out, lse, N, H, scale_log2, ds, thr, drop_key);
        CHB_LAUNCH_CHECK();
        return CHB_OK;
    }
#define CHB_FWD(NTP)                                                                                                      \
    do {                                                                                                                  \
        if (thr) hipLaunchKernelGGL((attn_fwd_kernel<NTP, true>), grid, block, 0, s, in, out, lse, N, H, scale_log2, ds, thr, drop_key); \
        else hipLaunchKernelGGL((attn_fwd_kernel<NTP, false>), grid, block, 0, s, in, out, lse, N, H, scale_log2, ds, thr, drop_key);    \
    } while (0)
    if (N <= 32) CHB_FWD(1);
    else if (N <= 64) CHB_FWD(2);
    else if (N <= 128) CHB_FWD(4);
    else CHB_FWD(7);
#undef CHB_FWD
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int B, int N, int H, int hd,
                      float drop_rate, uint32_t drop_key, float* dbias_qkv, float* dbias_ws, void* stream) {
    if (!qkv || !o || !d_o || !lse || !dqkv || B < 0 || N <= 0 || H <= 0 || drop_rate < 0.f || drop_rate >= 1.f) return CHB_EINVAL;
    if (hd != HD) return CHB_EUNSUPPORTED;
    if ((double)B * H * N * N >= 4294967296.0) return CHB_EUNSUPPORTED;
    if (B == 0) return CHB_OK;
    const float scale = 1.0f / sqrtf((float)hd);
    const float scale_log2 = 1.44269504088896340736f * scale;
    const float ds = 1.0f / (1.0f - drop_rate);
    const uint32_t thr = drop_rate > 0.f ? chb_drop_threshold(drop_rate) : 0u;
    const dim3 grid(B * H), block(512);
    hipStream_t s = (hipStream_t)stream;
    // N <= 224: one pass with the whole head resident in LDS.  Longer sequences (or CHB_ATTN_BWD_ALGO=2, used by the
    // parity tests to cross-check the two paths on the same input): dK/dV pass + dQ pass.
    const int bwd_algo = chb_option(CHB_OPT_ATTN_BWD_ALGO);
    const bool two_pass = N > 224 || bwd_algo == 2;
    if (two_pass) {
        const dim3 grid2(B * H, (N + 127) / 128);
        const bf16_t* a0 = (const bf16_t*)qkv;
        const bf16_t* a1 = (const bf16_t*)o;
        const bf16_t* a2 = (const bf16_t*)d_o;
        bf16_t* out = (bf16_t*)dqkv;
        if (thr) {
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<true>), grid2, block, 0, s, a0, a1, a2, lse, out, N, H, scale, scale_log2, ds, thr, drop_key, dbias_qkv);
            hipLaunchKernelGGL((attn_bwd_dq_kernel<true>), grid2, block, 0, s, a0, a1, a2, lse, out, N, H, scale, scale_log2, ds, thr, drop_key, dbias_qkv);
        } else {
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<false>), grid2, block, 0, s, a0, a1, a2, lse, out, N, H, scale, scale_log2, ds, thr, drop_key, dbias_qkv);
            hipLaunchKernelGGL((attn_bwd_dq_kernel<false>), grid2, block, 0, s, a0, a1, a2, lse, out, N, H, scale, scale_log2, ds, thr, drop_key, dbias_qkv);
        }
        CHB_LAUNCH_CHECK();
        return CHB_OK;
    }
    if (dbias_qkv && !dbias_ws) return CHB_EINVAL;      // the one-pass kernel writes per-batch-element rows, folded below
    float* db = dbias_qkv ? dbias_ws : nullptr;
    const int db_rows = dbias_qkv ? 1 : 0;
#define CHB_BWD_V(NTP, NW, DB)                                                                                                    \
    do {                                                                                                                         \
        const size_t lds = bwd_lds_bytes<NTP>();                                                                                 \
        static std::atomic<bool> attr_set{false};   /* once per instantiation: a driver call, not per launch (graph capture) */    \
        if (!attr_set.load(std::memory_order_acquire)) {                                                                         \
            if (hipFuncSetAttribute((const void*)attn_bwd_kernel<NTP, true, NW, DB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess || \
                hipFuncSetAttribute((const void*)attn_bwd_kernel<NTP, false, NW, DB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)  \
                return CHB_ELAUNCH;                                                                                              \
            attr_set.store(true, std::memory_order_release);                                                                     \
        }                                                                                                                        \
        if (thr) hipLaunchKernelGGL((attn_bwd_kernel<NTP, true, NW, DB>), grid, dim3(NW * 64), lds, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)d_o,  \
                           lse, (bf16_t*)dqkv, N, H, scale, scale_log2, ds, thr, drop_key, db);                                  \
        else hipLaunchKernelGGL((attn_bwd_kernel<NTP, false, NW, DB>), grid, dim3(NW * 64), lds, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)d_o,  \
                           lse, (bf16_t*)dqkv, N, H, scale, scale_log2, ds, thr, drop_key, db);                                  \
    } while (0)
    // the bias-gradient epilogue is its own instantiation: its four extra accumulators and the cross-wave fold cost the 16-wave
    // kernel (128 registers) 64-88 bytes of scratch per lane, without it the kernel does not spill
#define CHB_BWD(NTP, NW)                  \
    do {                                  \
        if (db) CHB_BWD_V(NTP, NW, true); \
        else CHB_BWD_V(NTP, NW, false);   \
    } while (0)
    if (N <= 32) CHB_BWD(1, 8);
    else if (N <= 64) CHB_BWD(2, 8);
    else if (N <= 128) CHB_BWD(4, 8);
    else if (bwd_algo == 1) CHB_BWD(7, 8);   // 8-wave variant kept for A/B timing
    else CHB_BWD(7, 16);
#undef CHB_BWD_V
#undef CHB_BWD
    if (db_rows) hipLaunchKernelGGL(dbias_reduce_kernel, dim3((3 * H * HD + 255) / 256, 16), dim3(256), 0, s, dbias_ws, B, 3 * H * HD, dbias_qkv);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

}  // extern "C"
