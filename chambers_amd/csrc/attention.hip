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
#include <mutex>

namespace {

constexpr int HD = 64;  // head dim (all ViT configs of the reference use 64)

__device__ __forceinline__ int swz_row(int r) { return (r >> 1) & 7; }          // row reads (ds_read_b128)
__device__ __forceinline__ int swz_trv(int r) { return ((r >> 1) & 3) << 1; }   // V image, transposed reads only
// images read BOTH ways (row fragments by ds_read_b128, transposed fragments by ds_read_b64_tr_b16): one 16-byte-chunk permutation
// that is conflict-free for the row reads (the 8 same-parity rows of a 16-lane-group pattern get 8 distinct chunks) and for the
// transposed reads (the 4 same-parity rows of the 8 rows a 32-lane half touches - rows r0..r0+7, or r0..r0+3 and r0+8..r0+11 - get 4
// distinct 32-byte chunk pairs).  v = (r >> 1) & 7 -> {0, 2, 4, 6, 5, 7, 1, 3}.
__device__ __forceinline__ int swz_dual(int r) {
    const int v = (r >> 1) & 7;
    return (((v + ((v >> 2) << 1)) & 3) << 1) | (v >> 2);
}
__device__ __forceinline__ bf16x8_t lds_row_frag_dual(const bf16_t* img, int r, int chunk) {
    return *reinterpret_cast<const bf16x8_t*>(img + r * 64 + ((chunk ^ swz_dual(r)) << 3));
}

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

__device__ __forceinline__ bf16x8_t lds_tr_frag_dual(const bf16_t* img, int ra, int rb, int c0, int i) {
    const int q = i >> 2, pp = i & 3;
    const int chunk = (c0 >> 3) + (pp >> 1);
    const int r0 = ra + q, r1 = rb + q;
    const bf16_t* a0 = img + r0 * 64 + ((chunk ^ swz_dual(r0)) << 3) + 4 * (pp & 1);
    const bf16_t* a1 = img + r1 * 64 + ((chunk ^ swz_dual(r1)) << 3) + 4 * (pp & 1);
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
        const uint32_t ebase = ((uint32_t)bh * (uint32_t)N + (uint32_t)min(q, N - 1)) * (uint32_t)((N + 3) & ~3);   // row stride Np4; B*H*N*Np4 < 2^32 (host)
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

// ------------------------------------------------------------------------------------ forward, whole head per workgroup
// N <= 256.  One workgroup = one (image, head): K and V of the head are brought into LDS ONCE by LDS-DMA (global_load_lds, no
// register round trip; the bank swizzle is applied to the per-lane SOURCE chunk) and the waves then run without any further
// barrier: wave w owns the 16-query tiles w, w + NW, ... (NW = ceil(tiles / 2): 7 waves for 197 tokens, so three workgroups
// share a CU - 160 KiB of LDS, 21 waves - and one's load phase runs under the others' arithmetic).  A query tile walks the keys in
// chunks of 64 with an online softmax (scores of 4 key tiles in 16 registers); key tiles past N cost nothing (wave-uniform
// skips).  Dropout: element index ((b*H+h)*N + q)*Np4 + k with the row stride Np4 = N rounded up to 4, so the 4 consecutive keys
// a lane holds per tile are the four 16-bit halves of exactly TWO hashes; the keep bits are also written out (8 bytes per
// query and lane group: drop_bits[((bh*N + q)*4 + g)*2 + (t >> 3)], bit 16*(r&1) + 8*(r>>1) + (t & 7) for key 16*t + 4*g + r) so that the
// backward pass tests a bit instead of hashing again.
// halves of x that are >= thr (both 16-bit, unsigned): bit 0 / bit 16 of the result.  x - (thr - 1) saturating at 0 is nonzero
// exactly for those halves; min(.., 1) makes it a flag.
__device__ __forceinline__ uint32_t chb_pk_ge_u16(uint32_t x, uint32_t thr) {
    const uint32_t tm1 = (thr - 1u) * 0x10001u;   // thr >= 1 (dropout rate > 0)
    uint32_t d;
    asm("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]" : "=&v"(d) : "v"(x), "v"(tm1));
    return d;
}

// (an inline-asm v_max3_f32 on MFMA results was tried for the row maximum and is WRONG: hipcc's hazard recognizer does not pad an asm
// statement that reads a register an MFMA is still writing - the s_nop it places in front of ordinary consumers is missing)
__device__ __forceinline__ void glds16_attn(const bf16_t* src, bf16_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_wave_base), 16, 0, 0);
}

struct FwdState {
    float& m_run;
    float& l_run;
    float4_t (&oacc)[4];
    uint32_t& bits_lo;
    uint32_t& bits_hi;
};

// one chunk (key tiles 4kc .. 4kc + nlive - 1) of one 16-query tile: scores, online-softmax update, dropout, P.V
template <bool FULL, bool DROP>
__device__ __forceinline__ void fwd_chunk(FwdState& st, const bf16_t* Ks, const bf16_t* Vs, const bf16x8_t& qf0, const bf16x8_t& qf1, int kc,
                                          int nlive, int N, int g, int i, uint32_t cbase, float scale_log2, uint32_t drop_thr,
                                          uint32_t drop_key) {
    float4_t s[4];
    float cmax = -INFINITY;
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        const int t = 4 * kc + tl;
        s[tl] = (float4_t){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (FULL || tl < nlive) {   // wave-uniform
            s[tl] = (float4_t){0.f, 0.f, 0.f, 0.f};
            s[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ks, 16 * t + i, g), qf0, s[tl], 0, 0, 0);
            s[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ks, 16 * t + i, 4 + g), qf1, s[tl], 0, 0, 0);
            if (!FULL && 16 * t + 16 > N) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s[tl][r] = (16 * t + 4 * g + r < N) ? s[tl][r] : -INFINITY;
            }
            cmax = fmaxf(cmax, fmaxf(fmaxf(s[tl][0], s[tl][1]), fmaxf(s[tl][2], s[tl][3])));
            if (!DROP) __builtin_amdgcn_sched_barrier(0);     // without the hash work to interleave, hipcc hoists all eight K fragment reads of
        }                                                     // a chunk in front of its first MFMA: +32 live registers, spills in the pipelined kernel
    }
    cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
    const float m_new = fmaxf(st.m_run, cmax);           // finite: every chunk holds at least one valid key
    const float corr = __builtin_amdgcn_exp2f((st.m_run - m_new) * scale_log2);   // first chunk: exp2(-inf) = 0
    st.m_run = m_new;
    const float mxs = m_new * scale_log2;
    st.l_run *= corr;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) st.oacc[dt] *= corr;
    // exponentials and the row sum (before dropout); dropout flags of a tile: the two 16-bit halves of a hash are compared at
    // once (chb_pk_ge_u16: bit 0 / 16), tile tl of the chunk at bit tl of each byte lane (0, 16, 8, 24 <-> r = 0..3).
    // Dead tiles of the last chunk are skipped (wave-uniform) and contribute exact zeros.
    uint32_t nib4 = 0u, fl[4][2];
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        fl[tl][0] = fl[tl][1] = 0u;
        if (FULL || tl < nlive) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[tl][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[tl][r], scale_log2, -mxs));
                st.l_run += s[tl][r];
            }
            if (DROP) {
                const uint32_t c0 = cbase + 8u * (uint32_t)(4 * kc + tl);
                const uint32_t h0 = chb_hash32(c0 ^ drop_key), h1 = chb_hash32((c0 + 1u) ^ drop_key);
                fl[tl][0] = chb_pk_ge_u16(h0, drop_thr);
                fl[tl][1] = chb_pk_ge_u16(h1, drop_thr);
                nib4 |= (fl[tl][0] | (fl[tl][1] << 8)) << tl;
            }
        } else {
            s[tl] = (float4_t){0.f, 0.f, 0.f, 0.f};
        }
    }
    // P^T fragments of the two key-tile pairs, bf16; the dropout mask is applied to the packed pairs (flag * 0xffff per half)
    uint4 pw[2];
#pragma unroll
    for (int ul = 0; ul < 2; ++ul) pw[ul] = __builtin_bit_cast(uint4, pack8(s[2 * ul], s[2 * ul + 1]));
    if (DROP) {
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            uint32_t* w = &pw[tl >> 1].x + 2 * (tl & 1);
            w[0] &= __umul24(fl[tl][0], 0xffffu);       // flags at bits 0 / 16 -> 0xffff per kept half; a 24-bit multiply is full rate, a
            w[1] &= __umul24(fl[tl][1], 0xffffu);       // 32-bit v_mul_lo_u32 a quarter of it (8 of them were 9 % of a chunk's vector cycles)
        }
        if (kc & 2) st.bits_hi |= nib4 << (4 * (kc & 1));   // wave-uniform
        else st.bits_lo |= nib4 << (4 * (kc & 1));
    }
    // O^T[d][q] += V^T[d][key] P^T[key][q]; k-slot (g, j): j<4 -> key 32u+4g+j, j>=4 -> key 32u+16+4g+(j-4)
#pragma unroll
    for (int ul = 0; ul < 2; ++ul) {
        const int u = 2 * kc + ul;
        if (FULL || 2 * ul < nlive) {   // wave-uniform; the second tile of the last pair may be dead: its P is 0, read a live V tile
            const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw[ul]);
            const int rb = (FULL || 2 * ul + 1 < nlive) ? 32 * u + 16 + 4 * g : 32 * u + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8_t vf = lds_tr_frag<true>(Vs, 32 * u + 4 * g, rb, 16 * dt, i);
                st.oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, st.oacc[dt], 0, 0, 0);
            }
        }
    }
}

template <bool DROP>
__global__ void __launch_bounds__(512, 4) attn_fwd_head_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse_out,
                                                            uint32_t* __restrict__ bits_out, int N, int H, int Np4, float scale_log2,
                                                            float drop_scale, uint32_t drop_thr, uint32_t drop_key) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int nkt = (N + 15) >> 4;                 // live key tiles == query tiles
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* Vs = Ks + nkt * 16 * HD;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = blockDim.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int Dm = H * HD;
    const int64_t D3 = 3 * (int64_t)Dm;
    const bf16_t* base = qkv + (int64_t)b * N * D3 + h * HD;

    // K / V images: one LDS-DMA instruction = 8 rows x 128 bytes; rows >= N re-read row N-1 (finite data; their scores are masked
    // and their probabilities are exactly 0)
    {
        const int r8 = lane >> 3, c = lane & 7;
        const int ninst = nkt * 2;
        for (int inst = wave; inst < ninst; inst += NW) {
            const int r = inst * 8 + r8;
            const int rs = min(r, N - 1);
            const bf16_t* kp = base + (int64_t)rs * D3 + Dm;
            glds16_attn(kp + ((c ^ swz_row(r)) << 3), Ks + inst * 512);
            glds16_attn(kp + Dm + ((c ^ swz_trv(r)) << 3), Vs + inst * 512);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int qt = wave; qt < nkt; qt += NW) {
        const int q0 = qt * 16;
        const int q = q0 + i, qc = min(q, N - 1);
        const bf16_t* qp = base + (int64_t)qc * D3;
        const bf16x8_t qf0 = *reinterpret_cast<const bf16x8_t*>(qp + g * 8);
        const bf16x8_t qf1 = *reinterpret_cast<const bf16x8_t*>(qp + 32 + g * 8);
        // hash counter of this lane's first key pair in tile 0: ((row * Np4) + 4g) / 2; tile t adds 8t
        const uint32_t cbase = ((((uint32_t)bh * (uint32_t)N + (uint32_t)qc) * (uint32_t)Np4) >> 1) + 2u * (uint32_t)g;
        float m_run = -INFINITY, l_run = 0.f;
        float4_t oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] = (float4_t){0.f, 0.f, 0.f, 0.f};
        uint32_t bits_lo = 0u, bits_hi = 0u;
        // chunks of 4 key tiles whose 64 keys all exist run a branch-free body (FULL: no liveness tests, no key mask); the last
        // chunk of the row takes the general body with its wave-uniform tile count
        const int nfull = N >> 6;
        const int nchunk = (nkt + 3) >> 2;
        FwdState st{m_run, l_run, oacc, bits_lo, bits_hi};
#pragma unroll 1
        for (int kc = 0; kc < nfull; ++kc)
            fwd_chunk<true, DROP>(st, Ks, Vs, qf0, qf1, kc, 4, N, g, i, cbase, scale_log2, drop_thr, drop_key);
        if (nfull < nchunk) fwd_chunk<false, DROP>(st, Ks, Vs, qf0, qf1, nfull, nkt - 4 * nfull, N, g, i, cbase, scale_log2, drop_thr, drop_key);
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
            if (DROP && bits_out) *reinterpret_cast<uint2*>(bits_out + (((int64_t)bh * N + q) * 4 + g) * 2) = make_uint2(bits_lo, bits_hi);
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
            const uint32_t un = (uint32_t)__builtin_amdgcn_readfirstlane((N + 3) & ~3);   // row stride Np4 of the mask index
            const uint32_t e0 = ((uint32_t)bh * (uint32_t)N + (uint32_t)(q0 + 4 * g)) * un;
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

// dropout keep factors for 4 consecutive elements e0 .. e0+3, e0 EVEN (row stride and key offset are multiples of 4): the 16-bit
// halves of two hashes
__device__ __forceinline__ void keep4_even(uint32_t e0, uint32_t drop_key, uint32_t drop_thr, float drop_scale, float out[4]) {
    const uint32_t c0 = e0 >> 1;
    const uint32_t h0 = chb_hash32(c0 ^ drop_key), h1 = chb_hash32((c0 + 1u) ^ drop_key);
    out[0] = ((h0 & 0xffffu) >= drop_thr) ? drop_scale : 0.f;
    out[1] = ((h0 >> 16) >= drop_thr) ? drop_scale : 0.f;
    out[2] = ((h1 & 0xffffu) >= drop_thr) ? drop_scale : 0.f;
    out[3] = ((h1 >> 16) >= drop_thr) ? drop_scale : 0.f;
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

// ------------------------------------------------------------------------------------ backward, lean one-pass kernel
// Same decomposition as attn_bwd_kernel with NW = 16 (wave w owns key tile w and keeps its K / V fragments and dK^T / dV^T in
// registers; dS crosses LDS once for dQ), rebuilt around the instruction count, which is what bounds it (PMC: the four waves of a
// SIMD issue ~76 % of the cycles, three quarters of that vector ALU):
//  * the dropout keep bits come from the forward pass (attn_fwd_head_kernel) instead of one hash per element: the head's bit rows
//    sit in LDS, a lane reads one word per query row and tests one lane-constant bit;
//  * key tiles and query sub-tiles that lie entirely past N are skipped (N = 197: 3 of 16 waves and half of the last query block);
//    the waves without a key tile are the ones that form dQ (phase B), off the critical path of the others;
//  * Q, dO and K images use one swizzle that is conflict-free for the row reads (ds_read_b128) AND for both transposed-read
//    patterns (swz_dual); dS is written transposed, [key][32 queries], as two 8-byte stores per tile instead of eight 2-byte ones,
//    with a piece / slot rotation that keeps those stores and the transposed reads of phase B conflict-free;
//  * 1/sqrt(hd) = 2^-3 is exact in bf16, so it multiplies dQ and dK when they are stored instead of every dS element;
//  * Q and K arrive by LDS-DMA, V and K fragments straight from global memory (the V image is gone: 28 KiB less LDS);
//    lse / delta are read as float4.
// workgroup barrier that orders LDS traffic only: every LDS operation of this wave has completed (lgkmcnt(0)), global stores and
// LDS-DMA stay in flight across it (__syncthreads() also waits vmcnt(0): a dQ store issued right before it would put the store's
// round trip to HBM on the critical path of every step)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


template <int NTP>
constexpr size_t bwd_head_lds_bytes() {
    return (size_t)(3 * 32 * NTP * HD + 2 * 32 * NTP * 32) * sizeof(bf16_t) + (size_t)2 * 32 * NTP * sizeof(float) + (size_t)32 * NTP * 10 * sizeof(uint32_t);
}

// byte offset inside one dS^T buffer of the 8-byte slot holding queries 16*qs + 4*gq .. +3 of key row `key`
__device__ __forceinline__ int dst_slot(int key, int qs, int gq) {
    return key * 64 + ((qs ^ ((key >> 3) & 1)) << 5) + (((gq + (key >> 1)) & 3) << 3);
}

template <int NTP, bool DROP, bool BITS>
__global__ void __launch_bounds__(1024, 4) attn_bwd_head_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o, const bf16_t* __restrict__ d_o,
                                                             const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                             const uint32_t* __restrict__ drop_bits, int N, int H, float scale,
                                                             float scale_log2, float drop_scale, uint32_t drop_thr, uint32_t drop_key, int dbg) {
    constexpr int NP = 32 * NTP, NTHR = 1024;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* Qs = Ks + NP * HD;
    bf16_t* Gs = Qs + NP * HD;                      // dO
    char* dST = reinterpret_cast<char*>(Gs + NP * HD);                  // [2][NP keys][64 bytes]
    float* lse2 = reinterpret_cast<float*>(dST + 2 * NP * 64);
    float* delta = lse2 + NP;
    uint32_t* bitsL = reinterpret_cast<uint32_t*>(delta + NP);         // [NP queries][10]: 4 g x 2 halves + 2 pad words (a lane group reads
                                                                       // rows 4 apart: 40 words, so the four groups use disjoint banks)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int Dm = H * HD;
    const int64_t D3 = 3 * (int64_t)Dm;
    const bf16_t* base = qkv + (int64_t)b * N * D3 + h * HD;
    const bf16_t* obase = o + (int64_t)b * N * Dm + h * HD;
    const bf16_t* gbase = d_o + (int64_t)b * N * Dm + h * HD;
    const int nkt = (N + 15) >> 4;                  // live 16-key tiles
    const int nqb = dbg == 1 ? 0 : (N + 31) >> 5;   // live 32-query blocks (dbg = 1: timing build of prologue + epilogue alone)
    const int t = wave;                             // this wave's key tile
    const bool tile_live = t < nkt;                 // wave-uniform
    const int key = 16 * t + i;

    // ---- prologue: Q and K images by LDS-DMA (rows >= N re-read row N-1: finite, and every use of them is masked)
    {
        const int r8 = lane >> 3, c = lane & 7;
        for (int inst = wave; inst < NP / 8; inst += 16) {
            const int r = inst * 8 + r8;
            const bf16_t* src = base + (int64_t)min(r, N - 1) * D3 + ((c ^ swz_dual(r)) << 3);
            glds16_attn(src, Qs + inst * 512);
            glds16_attn(src + Dm, Ks + inst * 512);
        }
    }
    // this wave's K / V row fragments (B operands of S and dP) straight from global memory
    bf16x8_t kfr[2], vfr[2];
    {
        const bf16_t* kp = base + (int64_t)min(key, N - 1) * D3 + Dm;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kfr[ks] = *reinterpret_cast<const bf16x8_t*>(kp + 32 * ks + 8 * g);
            vfr[ks] = *reinterpret_cast<const bf16x8_t*>(kp + Dm + 32 * ks + 8 * g);
        }
    }
    // dO image + delta[q] = sum_d dO*O (8 threads per row, one 16-byte chunk each); rows >= N are zero
    for (int id = tid; id < ((NP * 8 + NTHR - 1) & ~(NTHR - 1)); id += NTHR) {
        const int r = id >> 3, c = id & 7;
        uint4 gv = make_uint4(0, 0, 0, 0), ov = make_uint4(0, 0, 0, 0);
        if (r < N && r < NP) {
            gv = *reinterpret_cast<const uint4*>(gbase + (int64_t)r * Dm + c * 8);
            ov = *reinterpret_cast<const uint4*>(obase + (int64_t)r * Dm + c * 8);
        }
        if (r < NP) *reinterpret_cast<uint4*>(Gs + r * HD + ((c ^ swz_dual(r)) << 3)) = gv;
        float d = dot8_bf16(gv, ov);
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        if (c == 0 && r < NP) delta[r] = d;
    }
    for (int r = tid; r < NP; r += NTHR) lse2[r] = r < N ? lse[(int64_t)bh * N + r] * 1.44269504088896340736f : INFINITY;
    if (BITS) {
        const uint32_t* src = drop_bits + (int64_t)bh * N * 8;
        for (int id = tid; id < NP * 8; id += NTHR) bitsL[(id >> 3) * 10 + (id & 7)] = id < N * 8 ? src[id] : 0u;
    }
    for (int id = tid; id < 2 * NP * 4; id += NTHR) reinterpret_cast<uint4*>(dST)[id] = make_uint4(0, 0, 0, 0);   // rows of dead key tiles stay 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float4_t dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        dk[dt] = (float4_t){0.f, 0.f, 0.f, 0.f};
        dv[dt] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }
    // lane-constant LDS offsets (query blocks start at multiples of 32, so every swizzle depends on the lane only)
    const int lq = i >> 2, lpp = i & 3;
    int rowoff[2];            // row fragments of Q / dO: row (16 qs) + i, chunk 4 ks + g
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) rowoff[ks] = i * HD + (((4 * ks + g) ^ swz_dual(i)) << 3);
    int troff[4];             // transposed fragments of Q / dO: rows 4g + lq (and + 16), columns 16 dt + 4 lpp ..
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) troff[dt] = (4 * g + lq) * HD + (((2 * dt + (lpp >> 1)) ^ swz_dual(4 * g + lq)) << 3) + 4 * (lpp & 1);
    const int dst_w0 = dst_slot(key, 0, g), dst_w1 = dst_slot(key, 1, g);     // this lane's two dS^T stores (key row, query groups g / 4+g)
    const int bit_off = (4 * g) * 10 + (i >> 2) * 2 + (t >> 3);               // word of query row 4g (+ r*10) for this lane's key
    const int bit_pos = 16 * (i & 1) + 8 * ((i >> 1) & 1) + (t & 7);
    const uint32_t dsc_bits = __float_as_uint(drop_scale);
    // keys >= N of the partial tile: the score chain starts from -inf instead of 0, so p = exp2(-inf) = 0 without a select
    const float s0 = key < N ? 0.f : -INFINITY;
    const float4_t sinit = (float4_t){s0, s0, s0, s0};
    // Roles are fixed for the whole kernel and exclusive (N <= 224: at least two of the 16 waves own no live key tile): the waves
    // with a key tile run phase A of block `it`, the others form dQ of block it - 1 (phase B) from the dS^T tile published one
    // round earlier, between the same pair of barriers.  Two separate loops, so the register allocation of one role does not
    // carry the other's live state (dK / dV accumulators and K / V fragments on one side, 14 prefetched fragments on the other).
    const int nwork = 16 - nkt;
    const int widx = wave - nkt;
    // phase-B fragment addresses = lane constant + 32u rows: K^T rows 8g + lq (+4), d columns 16dt + 4lpp ..; dS^T rows likewise,
    // query columns 16qs + 4lpp .. (swizzle / piece / slot of row 32u + r equal those of row r)
    const int kra = 8 * g + lq, krb = kra + 4;
    int kboff[4][2], sboff[2][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        kboff[dt][0] = (kra * HD + (((2 * dt + (lpp >> 1)) ^ swz_dual(kra)) << 3) + 4 * (lpp & 1)) * 2;
        kboff[dt][1] = (krb * HD + (((2 * dt + (lpp >> 1)) ^ swz_dual(krb)) << 3) + 4 * (lpp & 1)) * 2;
    }
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        sboff[qs][0] = dst_slot(kra, qs, lpp);
        sboff[qs][1] = dst_slot(krb, qs, lpp);
    }
    if (tile_live) {
        for (int it = 0; it <= nqb; ++it) {
            const int q0 = 32 * it;
            if (it < nqb) {
                // ---- phase A: this wave's 16 keys against the (up to) 32 queries of the block
                char* dSb = dST + (it & 1) * NP * 64;
                const bool two = q0 + 16 < N;                         // second 16-query sub-tile has live queries (wave-uniform)
                float4_t pd[2], ds[2];
#pragma unroll
                for (int qs = 0; qs < 2; ++qs) {
                    pd[qs] = (float4_t){0.f, 0.f, 0.f, 0.f};
                    ds[qs] = (float4_t){0.f, 0.f, 0.f, 0.f};
                    if (qs == 0 || two) {
                        const bf16_t* qrow = Qs + (q0 + 16 * qs) * HD;
                        const bf16_t* grow = Gs + (q0 + 16 * qs) * HD;
                        const float4_t l2 = *reinterpret_cast<const float4_t*>(lse2 + q0 + 16 * qs + 4 * g);
                        const float4_t dl = *reinterpret_cast<const float4_t*>(delta + q0 + 16 * qs + 4 * g);
                        uint32_t bw[4] = {0u, 0u, 0u, 0u};
                        if (BITS) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) bw[r] = bitsL[(q0 + 16 * qs + r) * 10 + bit_off];
                        }
                        float4_t sv = sinit, dp = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            // D[row = query 4g+r][col = key i]
                            sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(qrow + rowoff[ks]), kfr[ks], sv, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(grow + rowoff[ks]), vfr[ks], dp, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], scale_log2, -l2[r]));    // pad queries: lse = +inf -> 0
                            float keepc = 1.0f;
                            if (DROP) {
                                if (BITS) {
                                    const int32_t m = __builtin_amdgcn_sbfe((int32_t)bw[r], bit_pos, 1);      // 0 or -1
                                    keepc = __uint_as_float((uint32_t)m & dsc_bits);
                                } else {
                                    const uint32_t e = ((uint32_t)bh * (uint32_t)N + (uint32_t)min(q0 + 16 * qs + 4 * g + r, N - 1)) * (uint32_t)((N + 3) & ~3) +
                                                       (uint32_t)min(key, N - 1);
                                    const uint32_t hsh = chb_hash32((e >> 1) ^ drop_key);
                                    const uint32_t u = (e & 1u) ? (hsh >> 16) : (hsh & 0xffffu);
                                    keepc = (u >= drop_thr) ? drop_scale : 0.f;
                                }
                            }
                            pd[qs][r] = p * keepc;                                  // dropped probabilities (for dV)
                            ds[qs][r] = p * __builtin_fmaf(dp[r], keepc, -dl[r]);   // d(scores) * sqrt(hd): 2^-3 goes onto dQ / dK at the store
                        }
                    }
                }
                // contraction over the 32 queries: k-slot (g, j): j<4 -> q0+4g+j, j>=4 -> q0+16+4g+(j-4)
                const bf16x8_t pf = pack8(pd[0], pd[1]);
                const bf16x8_t sf = pack8(ds[0], ds[1]);
                const bf16_t* gt0 = Gs + q0 * HD;
                const bf16_t* qt0 = Qs + q0 * HD;
                const int second = two ? 16 * HD : 0;       // dead second sub-tile: its P / dS are 0, read the live rows again
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8_t gt = lds_tr_frag_at(gt0 + troff[dt], gt0 + second + troff[dt]);
                    const bf16x8_t qt = lds_tr_frag_at(qt0 + troff[dt], qt0 + second + troff[dt]);
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt, pf, dv[dt], 0, 0, 0);  // dV^T[d][key]
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, sf, dk[dt], 0, 0, 0);  // dK^T[d][key]
                }
                // dS^T tile -> LDS [key][query]: the packed fragment IS the two 8-byte rows (queries 4g.., 16+4g..)
                const uint4 sw = __builtin_bit_cast(uint4, sf);
                *reinterpret_cast<uint2*>(dSb + dst_w0) = make_uint2(sw.x, sw.y);
                *reinterpret_cast<uint2*>(dSb + dst_w1) = make_uint2(sw.z, sw.w);
            }
            lds_barrier();   // dS^T of block `it` is published; buffer (it - 1) & 1 is free for block it + 1
        }
    } else {
        for (int it = 0; it <= nqb; ++it) {
            // ---- phase B (block it - 1): dQ[q][d] = sum_key dS[q][key] K[key][d], one 16 x 16 tile (qs, dt) per task
            if (it > 0) {
                const int qb0 = 32 * (it - 1);
                const char* dSb = dST + ((it - 1) & 1) * NP * 64;
                for (int task = widx; task < 8; task += nwork) {
                    const int qs = task >> 2, dtw = task & 3;
                    if (qb0 + 16 * qs < N) {   // wave-uniform
                        // A[row d][k = key 32u+8g+j] = K^T: transposed read of the K image; B[k = key][col = query] = dS^T: transposed read
                        // of the [key][query] image (lane 4q'+p supplies row 8g+q' (+4), queries 16qs + 4p ..).  All fragment reads of the
                        // task are issued before its MFMA chain (compile-time trip count: pairs past the last live tile read zero dS rows).
                        bf16x8_t kt[NTP], sb[NTP];
                        const char* kp0 = reinterpret_cast<const char*>(Ks) + (dtw == 0 ? kboff[0][0] : dtw == 1 ? kboff[1][0] : dtw == 2 ? kboff[2][0] : kboff[3][0]);
                        const char* kp1 = reinterpret_cast<const char*>(Ks) + (dtw == 0 ? kboff[0][1] : dtw == 1 ? kboff[1][1] : dtw == 2 ? kboff[2][1] : kboff[3][1]);
                        const char* sp0 = dSb + (qs ? sboff[1][0] : sboff[0][0]);
                        const char* sp1 = dSb + (qs ? sboff[1][1] : sboff[0][1]);
#pragma unroll
                        for (int u = 0; u < NTP; ++u) {
                            kt[u] = lds_tr_frag_at(reinterpret_cast<const bf16_t*>(kp0 + u * 32 * HD * 2), reinterpret_cast<const bf16_t*>(kp1 + u * 32 * HD * 2));
                            sb[u] = lds_tr_frag_at(reinterpret_cast<const bf16_t*>(sp0 + u * 32 * 64), reinterpret_cast<const bf16_t*>(sp1 + u * 32 * 64));
                        }
                        float4_t dq = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int u = 0; u < NTP; ++u) dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt[u], sb[u], dq, 0, 0, 0);
                        // D[row = d 4g+r][col = query i]
                        const int q = qb0 + 16 * qs + i;
                        if (q < N) {
                            uint2 w;
                            w.x = pack_bf16x2(dq[0] * scale, dq[1] * scale);
                            w.y = pack_bf16x2(dq[2] * scale, dq[3] * scale);
                            *reinterpret_cast<uint2*>(dqkv + ((int64_t)b * N + q) * D3 + h * HD + 16 * dtw + 4 * g) = w;
                        }
                    }
                }
            }
            lds_barrier();
        }
    }
    // dK^T / dV^T accumulators: lane (g,i) reg r = [d = 16dt + 4g + r][key = 16t + i]
    if (tile_live && key < N) {
        bf16_t* kp = dqkv + ((int64_t)b * N + key) * D3 + Dm + h * HD;
        bf16_t* vp = kp + Dm;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            uint2 w;
            w.x = pack_bf16x2(dk[dt][0] * scale, dk[dt][1] * scale);
            w.y = pack_bf16x2(dk[dt][2] * scale, dk[dt][3] * scale);
            *reinterpret_cast<uint2*>(kp + 16 * dt + 4 * g) = w;
            w.x = pack_bf16x2(dv[dt][0], dv[dt][1]);
            w.y = pack_bf16x2(dv[dt][2], dv[dt][3]);
            *reinterpret_cast<uint2*>(vp + 16 * dt + 4 * g) = w;
        }
    }
}

// ------------------------------------------------------------------------------------ backward, persistent pipelined kernel
// 193 <= N <= 208 (ViT at 224^2 / patch 16 with one or two special tokens: 13 key tiles, 7 query blocks).  Same arithmetic as
// attn_bwd_head_kernel in the same order - the outputs are bit-identical - but the workgroup is PERSISTENT (one per CU, walking
// heads blockIdx.x, blockIdx.x + gridDim.x, ...) and nothing of a head is loaded "up front": the lean kernel spends half of its time
// (tools/attn_ablate.py: 0.28 of 0.56 ms) moving a head's 200 KB with the CU otherwise idle, because one 16-wave workgroup is all a
// CU holds (128 registers x 16 waves, 125 KiB of LDS) and its prologue / epilogue cannot overlap another head's arithmetic.
// Here the unit of work is a STEP = one 32-query block of one head, and the steps of all the workgroup's heads form one stream:
//  * Q / dO / keep-bit rows of a block are only read during that block's step, so they live in a ring of RING = 4 slots (9 KiB
//    each) instead of whole-head images; wave 15 (the PRODUCER) issues the LDS-DMA pieces of step p + 3 at step p and - one step
//    later, through registers - the O / dO rows whose dot products are delta = rowsum(dO * O), plus lse;
//  * K (both dS.K operands of phase B and the row fragments of phase A) and V are whole-head images: K double-buffered (phase B of a
//    head's last block runs during the first step of the next head), V single (read once, at a head's first step); waves 13 / 14
//    run phase B one step behind and issue the next head's K / V pieces during steps 1 .. 4 of the current head;
//  * waves 0 .. 12 own a key tile each (phase A) exactly as in the lean kernel; dK / dV leave with fire-and-forget stores at a head's
//    last step; one LDS-only barrier per step, none around head boundaries.
// Ordering (loads, stores and LDS-DMA of a wave retire in issue order): only the producer issues loads, and it consumes its register
// loads one barrier after issuing them - that data dependence is the only wait, see the producer loop.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
__device__ __forceinline__ uint32_t lds_addr32(const void* p) { return (uint32_t)(uintptr_t)LDS_PTR(p); }
// LDS-DMA piece (64 lanes x 16 bytes -> 1 KiB at lds_wave_base) as inline asm: hipcc's wait-count pass must not know these are
// LDS-DMA.  Measured on this kernel's ISA: with the builtin in the producer's loop, the pass put an s_waitcnt vmcnt(0) in front of
// the first transposed LDS read / dS^T store of EVERY iteration of the phase A and phase B loops too (loops of other waves, which no
// path connects to the producer's) - each dQ / dK / dV store would then be waited for, round trip to HBM included, one step later.
// m0 (the LDS base of the piece) is written inside the asm and not listed as a clobber (hipcc warns that it is a reserved register;
// it never holds a compiler value here: gfx950 DS instructions take no m0 and the kernel uses no builtin LDS-DMA, movrel or sendmsg -
// tools/check_pipe_isa.py checks that every m0 write in the kernel's ISA is one of these asm statements).
__device__ __forceinline__ void glds16_pipe(const void* src, const void* lds_wave_base) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_addr32(lds_wave_base));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory");   // m0 is reserved (not allocatable): the asm owns it, see the note above
}

// the same with a wave-uniform base in scalar registers and a 32-bit byte offset per lane: no 64-bit vector arithmetic per piece (the
// producer wave's address chains were a third of its issue time)
__device__ __forceinline__ uint64_t uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ __forceinline__ void glds16_pipe_s(const void* sbase, uint32_t voff, const void* lds_wave_base) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_addr32(lds_wave_base));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(uniform_ptr(sbase)), "s"(m0v) : "memory");
}
__device__ __forceinline__ void glds4_pipe_s(const void* sbase, uint32_t voff, const void* lds_wave_base) {      // 64 lanes x 4 bytes
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_addr32(lds_wave_base));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(uniform_ptr(sbase)), "s"(m0v) : "memory");
}

namespace pipe {
constexpr int NKT = 13, NQB = 7, NP = 224, RING = 4, ORING = 3;
constexpr int SLOT_ELEMS = 32 * HD;                              // bf16 elements of one Q / dO / O ring slot (4 KiB)
constexpr size_t OFF_Q = 0;
constexpr size_t OFF_G = OFF_Q + (size_t)RING * SLOT_ELEMS * 2;
constexpr size_t OFF_O = OFF_G + (size_t)RING * SLOT_ELEMS * 2;
constexpr size_t OFF_BITS = OFF_O + (size_t)ORING * SLOT_ELEMS * 2;
constexpr size_t OFF_LSE = OFF_BITS + (size_t)RING * 1024;       // 256 bytes a slot: raw lse by LDS-DMA (64 lanes x 4 bytes), fixed up in place
constexpr size_t OFF_DELTA = OFF_LSE + (size_t)RING * 256;
constexpr size_t OFF_K = OFF_DELTA + (size_t)RING * 128;
constexpr size_t OFF_V = OFF_K + (size_t)2 * NP * HD * 2;
constexpr size_t OFF_DST = OFF_V + (size_t)NKT * 16 * HD * 2;
constexpr size_t OFF_HEADS = OFF_DST + (size_t)2 * NP * 64;
constexpr size_t LDS_BYTES = OFF_HEADS + 16;                     // 163,344 of the CU's 163,840 bytes
static_assert(LDS_BYTES <= 160 * 1024, "the pipelined attention backward needs the whole LDS of a CU, not more");
}  // namespace pipe

// head counters of the persistent kernel: one slot per launch in flight (launches of different streams never share one); a launch
// leaves its slot at zero
constexpr int PIPE_CTR_SLOTS = 32;
__device__ unsigned int g_pipe_head_ctr[PIPE_CTR_SLOTS];

template <bool DROP>
__global__ void __launch_bounds__(1024, 4) attn_bwd_pipe_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o, const bf16_t* __restrict__ d_o,
                                                             const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                             const uint32_t* __restrict__ drop_bits, unsigned int* __restrict__ head_ctr, int N, int H, int total_heads, float scale,
                                                             float scale_log2, float drop_scale, int dbg) {
    const int dbg_period = dbg & 1023;          // timing experiments (tools/attn_pipe_check.py): results are WRONG when dbg != 0
    const bool dbg_no_b = dbg & 1024, dbg_no_prod = dbg & 2048, dbg_no_acc = dbg & 4096, dbg_no_soft = dbg & 8192;
    using namespace pipe;
    constexpr int NTP = NP / 32;
    constexpr int NDMA = DROP ? 14 : 13;           // producer pieces per step: Q 4, dO 4, O 4, lse 1, keep bits 1
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* ringQ = reinterpret_cast<bf16_t*>(smem_raw + OFF_Q);
    bf16_t* ringG = reinterpret_cast<bf16_t*>(smem_raw + OFF_G);
    bf16_t* ringO = reinterpret_cast<bf16_t*>(smem_raw + OFF_O);
    uint32_t* ringBits = reinterpret_cast<uint32_t*>(smem_raw + OFF_BITS);
    float* ringLse = reinterpret_cast<float*>(smem_raw + OFF_LSE);
    float* ringDelta = reinterpret_cast<float*>(smem_raw + OFF_DELTA);
    bf16_t* Kimg = reinterpret_cast<bf16_t*>(smem_raw + OFF_K);       // [2][NP][64]
    bf16_t* Vimg = reinterpret_cast<bf16_t*>(smem_raw + OFF_V);       // [16 NKT][64]
    char* dST = smem_raw + OFF_DST;                                    // [2][NP keys][64 bytes]
    int* heads = reinterpret_cast<int*>(smem_raw + OFF_HEADS);   // [4]: head of the k-th turn of this workgroup at [k & 3], -1 = none (plain LDS
                                                                 // accesses: every barrier of this kernel is a compiler memory barrier too)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int Dm = H * HD;
    const int64_t D3 = 3 * (int64_t)Dm;
    // Heads are CLAIMED, not assigned: a workgroup draws its next head from a device counter (ticket = head index), two heads ahead of
    // the one it is working on, so that a workgroup that starts late - its CU was held by another kernel, e.g. a collective running
    // beside this launch - simply draws fewer heads instead of stretching the launch by its whole static share.  Wave 13 draws: two
    // tickets in the prologue, one more at the first step of every head it has (exactly two tickets of every workgroup come back
    // >= total_heads; the holder of the last ticket of the launch, total_heads + 2 gridDim.x - 1, puts the counter back to zero).
    auto claim = [&]() -> int {            // wave 13, uniform: one ticket
        unsigned t = 0;
        if (lane == 0) {
            t = atomicAdd(head_ctr, 1u);
            if (t == (unsigned)total_heads + 2u * gridDim.x - 1u) atomicExch(head_ctr, 0u);
        }
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        return t < (unsigned)total_heads ? (int)t : -1;
    };
    const int r8 = lane >> 3, c8 = lane & 7;

    // one K or V image piece = 8 rows x 128 bytes (rows >= N re-read row N - 1: finite, and every use of them is masked)
    auto kv_piece = [&](int bh, int which, int inst, bf16_t* img) {
        const int b = bh / H, h = bh - b * H;
        const int r = inst * 8 + r8;
        const bf16_t* sbase = qkv + (int64_t)b * N * D3 + (int64_t)which * Dm + h * HD;
        const uint32_t voff = (__umul24((uint32_t)min(r, N - 1), (uint32_t)(3 * Dm)) + (uint32_t)((c8 ^ swz_dual(r)) << 3)) * 2u;
        glds16_pipe_s(sbase, voff, img + inst * 512);
    };

    // ---- prologue: the first two heads; zero dS^T (rows of dead key tiles stay zero for good); first head's K / V; the producer's first
    // three steps
    if (wave == 13) {
        const int h0 = claim();
        const int h1 = claim();
        if (lane == 0) {
            heads[0] = h0;
            heads[1] = h1;
            heads[2] = -1;
            heads[3] = -1;
        }
    }
    __syncthreads();
    const int head0 = heads[0];
    if (head0 < 0) return;                 // every head was taken before this workgroup started (uniform over the workgroup)
    for (int id = tid; id < 2 * NP * 4; id += 1024) reinterpret_cast<uint4*>(dST)[id] = make_uint4(0, 0, 0, 0);
    if (wave == 13) {
        for (int inst = 0; inst < NP / 8; ++inst) kv_piece(head0, 1, inst, Kimg);
    } else if (wave == 14) {
        for (int inst = 0; inst < NKT * 2; ++inst) kv_piece(head0, 2, inst, Vimg);
    }

    // ---- producer pieces.  Everything the producer moves goes global -> LDS by LDS-DMA (inline asm, see glds16_pipe); delta and the
    // scaled lse of a step are formed from LDS two iterations after its pieces were issued (asm LDS reads: a DS instruction hipcc can
    // see behind a pending LDS-DMA gets a vmcnt(0) in front), so no iteration waits for a round trip to HBM it has just started.
    auto step_head = [&](int tp, int& bh, int& q0) {       // step index -> (head, first query); a step past this workgroup's last head
        const int k = tp / NQB;                            // repeats the last real step (its pieces land in slots nobody reads)
        bh = heads[k & 3];
        q0 = 32 * (tp - k * NQB);
        if (bh < 0) {
            bh = heads[(k - 1) & 3];
            q0 = 32 * (NQB - 1);
        }
    };
    auto prod_dma = [&](int tp, int islot) {               // NDMA pieces of step tp -> ring slots of stream position islot
        int bh, q0;
        step_head(tp, bh, q0);
        const int b = bh / H, h = bh - b * H;
        const int slot = islot & (RING - 1), oslot = islot % ORING;
        const int rmax = N - 1 - q0;                       // rows past the last token re-read it (only a head's last block has any)
        const int64_t tok = (int64_t)b * N + q0;
        const bf16_t* qb = qkv + tok * D3 + h * HD;
        const bf16_t* gb = d_o + tok * Dm + h * HD;
        const bf16_t* ob = o + tok * Dm + h * HD;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int r = 8 * pc + r8;
            const uint32_t t = __umul24((uint32_t)min(r, rmax), (uint32_t)Dm);
            const uint32_t sw = (uint32_t)((c8 ^ swz_dual(r)) << 3);
            if (!(dbg & 16384)) glds16_pipe_s(qb, (3u * t + sw) * 2u, ringQ + slot * SLOT_ELEMS + pc * 512);     // (timing experiment: no Q pieces)
            glds16_pipe_s(gb, (t + sw) * 2u, ringG + slot * SLOT_ELEMS + pc * 512);
            glds16_pipe_s(ob, (t + sw) * 2u, ringO + oslot * SLOT_ELEMS + pc * 512);
        }
        glds4_pipe_s(lse + (int64_t)bh * N + q0, (uint32_t)min(lane & 31, rmax) * 4u, ringLse + slot * 64);
        if (DROP) {
            // LDS row s of the slot holds query row rho(s): rows 4 apart (the four lane groups of a fragment) sit 8 words apart
            const int s = lane >> 1, half = lane & 1;
            const int rho = (s & 0x18) | ((s & 1) << 2) | ((s >> 1) & 3);
            glds16_pipe_s(drop_bits + ((int64_t)bh * N + q0) * 8, (uint32_t)(min(rho, rmax) * 8 + 4 * half) * 4u, ringBits + slot * 256);
        }
    };
    auto prod_aux = [&](int tp, int islot) {               // delta / scaled lse of step tp from its landed dO / O / lse pieces
        int bh, q0;
        step_head(tp, bh, q0);
        const int slot = islot & (RING - 1), oslot = islot % ORING;
        u32x4_t gv[4], ov[4];
        float lraw;
        // lane (r8, c8) holds chunk c8 of rows r8 + 8 k4, as in the lean kernel's prologue (same sums in the same order)
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int r = r8 + 8 * k4;
            const uint32_t off = (uint32_t)((r * HD + ((c8 ^ swz_dual(r)) << 3)) * 2);
            asm volatile("ds_read_b128 %0, %1" : "=v"(gv[k4]) : "v"(lds_addr32(ringG + slot * SLOT_ELEMS) + off));
            asm volatile("ds_read_b128 %0, %1" : "=v"(ov[k4]) : "v"(lds_addr32(ringO + oslot * SLOT_ELEMS) + off));
        }
        asm volatile("ds_read_b32 %0, %1" : "=v"(lraw) : "v"(lds_addr32(ringLse + slot * 64 + (lane & 31))));
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]), "+v"(ov[0]), "+v"(ov[1]), "+v"(ov[2]), "+v"(ov[3]), "+v"(lraw)
                     :
                     : "memory");
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const bool live = q0 + r8 + 8 * k4 < N;
            float d = live ? dot8_bf16(make_uint4(gv[k4][0], gv[k4][1], gv[k4][2], gv[k4][3]), make_uint4(ov[k4][0], ov[k4][1], ov[k4][2], ov[k4][3])) : 0.f;
            // the 8 lanes of a row: lane ^ 1, lane ^ 2 (quad permutes), then the mirrored lane of the 8-group (it sits in the other quad,
            // whose four lanes all hold that quad's sum): the same three additions as __shfl_xor 1 / 2 / 4, as DPP moves instead of
            // ds_bpermute round trips (12 dependent LDS-crossbar trips an iteration were a third of the producer's time)
            d += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d), 0xB1, 0xF, 0xF, true));
            d += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d), 0x4E, 0xF, 0xF, true));
            d += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d), 0x141, 0xF, 0xF, true));
            if (c8 == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr32(ringDelta + slot * 32 + r8 + 8 * k4)), "v"(d) : "memory");
        }
        const float l2 = (q0 + lane < N) ? lraw * 1.44269504088896340736f : INFINITY;
        if (lane < 32) asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr32(ringLse + slot * 64 + lane)), "v"(l2) : "memory");
    };
    if (wave == 15) {
        prod_dma(0, 0);
        prod_dma(1, 1);
        prod_dma(2, 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 15) prod_aux(0, 0);
    __syncthreads();

    if (wave < NKT) {
        // ================================================================ phase A: this wave's 16 keys, one 32-query block a step
        const int t = wave;
        const int key = 16 * t + i;
        const int lq = i >> 2, lpp = i & 3;
        int rowoff[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) rowoff[ks] = i * HD + (((4 * ks + g) ^ swz_dual(i)) << 3);
        int troff[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) troff[dt] = (4 * g + lq) * HD + (((2 * dt + (lpp >> 1)) ^ swz_dual(4 * g + lq)) << 3) + 4 * (lpp & 1);
        const int dst_w0 = dst_slot(key, 0, g), dst_w1 = dst_slot(key, 1, g);
        const int bit_off = (8 * (g >> 1) + (g & 1)) * 8 + (i >> 2) * 2 + (t >> 3);       // + 128 qs + 16 r
        const int bit_pos = 16 * (i & 1) + 8 * ((i >> 1) & 1) + (t & 7);
        const uint32_t dsc_bits = __float_as_uint(drop_scale);
        const float s0 = key < N ? 0.f : -INFINITY;
        const float4_t sinit = (float4_t){s0, s0, s0, s0};
        float4_t dk[4], dv[4];
        bf16x8_t kfr[2], vfr[2];
        int k = 0, j = 0, bh = head0;
        for (int p = 0;; ++p) {
            if (j == 0 && p) bh = heads[k & 3];
            if (bh >= 0) {
                if (j == 0) {
                    const bf16_t* Kc = Kimg + (k & 1) * NP * HD;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        kfr[ks] = lds_row_frag_dual(Kc, key, 4 * ks + g);
                        vfr[ks] = lds_row_frag_dual(Vimg, key, 4 * ks + g);
                    }
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        dk[dt] = (float4_t){0.f, 0.f, 0.f, 0.f};
                        dv[dt] = (float4_t){0.f, 0.f, 0.f, 0.f};
                    }
                }
                const int slot = p & (RING - 1);
                const int q0 = 32 * j;
                const bf16_t* Qs = ringQ + slot * SLOT_ELEMS;
                const bf16_t* Gs = ringG + slot * SLOT_ELEMS;
                char* dSb = dST + (p & 1) * NP * 64;
                const bool two = q0 + 16 < N;
                float4_t pd[2], ds[2];
#pragma unroll
                for (int qs = 0; qs < 2; ++qs) {
                    pd[qs] = (float4_t){0.f, 0.f, 0.f, 0.f};
                    ds[qs] = (float4_t){0.f, 0.f, 0.f, 0.f};
                    if (qs == 0 || two) {
                        const bf16_t* qrow = Qs + (16 * qs) * HD;
                        const bf16_t* grow = Gs + (16 * qs) * HD;
                        const float4_t l2 = *reinterpret_cast<const float4_t*>(ringLse + slot * 64 + 16 * qs + 4 * g);
                        const float4_t dl = *reinterpret_cast<const float4_t*>(ringDelta + slot * 32 + 16 * qs + 4 * g);
                        uint32_t bw[4] = {0u, 0u, 0u, 0u};
                        if (DROP) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) bw[r] = ringBits[slot * 256 + 128 * qs + 16 * r + bit_off];
                        }
                        float4_t sv = sinit, dp = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(qrow + rowoff[ks]), kfr[ks], sv, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(grow + rowoff[ks]), vfr[ks], dp, 0, 0, 0);
                        }
                        if (dbg_no_soft) {
                            pd[qs] = sv;
                            ds[qs] = dp;
                        } else
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], scale_log2, -l2[r]));
                            float keepc = 1.0f;
                            if (DROP) {
                                const int32_t m = __builtin_amdgcn_sbfe((int32_t)bw[r], bit_pos, 1);
                                keepc = __uint_as_float((uint32_t)m & dsc_bits);
                            }
                            pd[qs][r] = pr * keepc;
                            ds[qs][r] = pr * __builtin_fmaf(dp[r], keepc, -dl[r]);
                        }
                    }
                }
                const bf16x8_t pf = pack8(pd[0], pd[1]);
                const bf16x8_t sf = pack8(ds[0], ds[1]);
                const int second = two ? 16 * HD : 0;
                if (!dbg_no_acc)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8_t gt = lds_tr_frag_at(Gs + troff[dt], Gs + second + troff[dt]);
                    const bf16x8_t qt = lds_tr_frag_at(Qs + troff[dt], Qs + second + troff[dt]);
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt, pf, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, sf, dk[dt], 0, 0, 0);
                }
                const uint4 sw = __builtin_bit_cast(uint4, sf);
                *reinterpret_cast<uint2*>(dSb + dst_w0) = make_uint2(sw.x, sw.y);
                *reinterpret_cast<uint2*>(dSb + dst_w1) = make_uint2(sw.z, sw.w);
                if (j == NQB - 1 && key < N) {
                    // the head is done for this key tile: dK^T / dV^T, lane (g,i) reg r = [d = 16dt + 4g + r][key]
                    const int b = bh / H, h = bh - b * H;
                    bf16_t* kp = dqkv + ((int64_t)b * N + key) * D3 + Dm + h * HD;
                    bf16_t* vp = kp + Dm;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        uint2 w;
                        w.x = pack_bf16x2(dk[dt][0] * scale, dk[dt][1] * scale);
                        w.y = pack_bf16x2(dk[dt][2] * scale, dk[dt][3] * scale);
                        *reinterpret_cast<uint2*>(kp + 16 * dt + 4 * g) = w;
                        w.x = pack_bf16x2(dv[dt][0], dv[dt][1]);
                        w.y = pack_bf16x2(dv[dt][2], dv[dt][3]);
                        *reinterpret_cast<uint2*>(vp + 16 * dt + 4 * g) = w;
                    }
                }
            }
            if (dbg_period <= 1 || p % dbg_period == 0) lds_barrier();   // dbg_period > 1: timing experiment only (results WRONG)
            if (bh < 0) break;            // the iteration behind this workgroup's last step: phase B finishes the last block in it
            if (++j == NQB) {
                j = 0;
                ++k;
            }
        }
    } else if (wave < 15) {
        // ================================================================ phase B (one step behind) + the next head's K / V pieces
        const int widx = wave - NKT;
        const int lq = i >> 2, lpp = i & 3;
        const int kra = 8 * g + lq, krb = kra + 4;
        int kboff[4][2], sboff[2][2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            kboff[dt][0] = (kra * HD + (((2 * dt + (lpp >> 1)) ^ swz_dual(kra)) << 3) + 4 * (lpp & 1)) * 2;
            kboff[dt][1] = (krb * HD + (((2 * dt + (lpp >> 1)) ^ swz_dual(krb)) << 3) + 4 * (lpp & 1)) * 2;
        }
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            sboff[qs][0] = dst_slot(kra, qs, lpp);
            sboff[qs][1] = dst_slot(krb, qs, lpp);
        }
        int k = 0, j = 0, pk = 0, pj = 0, bh = head0, pbh = -1, nxt = -1;
        bf16x8_t kt[2][NTP];
        for (int p = 0;; ++p) {
            if (j == 0) {
                if (p) bh = heads[k & 3];
                if (bh >= 0 && wave == 13) {          // this workgroup's head after next (first read at step 1 of the next head)
                    const int h2 = claim();
                    if (lane == 0) heads[(k + 2) & 3] = h2;
                }
            }
            if (j == 1) nxt = heads[(k + 1) & 3];     // drawn at the first step of the previous head (or in the prologue)
            // the next head's K (wave 13) / V (wave 14) pieces go out at steps 1 .. 4 of a head, 7 a step (a piece costs its issuer
            // 60 - 180 cycles: the producer alone was issue-bound); at the top of step 6 at least this wave's 8 dQ stores of steps 4
            // and 5 (blocks 3 and 4 are full for N >= 193) have been issued behind the last piece, so "at most 4 outstanding" means
            // every piece has landed - the barrier that ends step 6 publishes the images
            if (bh >= 0 && j >= 1 && j <= 4 && nxt >= 0 && !dbg_no_prod) {
                const int first = 7 * (j - 1);
                if (wave == 13) {
                    bf16_t* img = Kimg + ((k + 1) & 1) * NP * HD;
#pragma unroll
                    for (int u = 0; u < 7; ++u) kv_piece(nxt, 1, first + u, img);                            // 28 pieces
                } else {
#pragma unroll
                    for (int u = 0; u < 7; ++u) kv_piece(nxt, 2, min(first + u, NKT * 2 - 1), Vimg);          // 26 pieces (the last one three times)
                }
            }
            if (bh >= 0 && j == NQB - 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (p > 0 && !dbg_no_b) {
                const int b = pbh / H, h = pbh - b * H;
                const int qb0 = 32 * pj;
                const char* dSb = dST + ((p - 1) & 1) * NP * 64;
                if (pj == 0) {
                    // first block of a head: this wave's K^T fragments (A[row d][k = key 32u + 8g + j], d tiles widx and widx + 2) stay in
                    // registers for the head's seven blocks - the lean kernel re-read them from the K image for every task
                    const char* Kc = reinterpret_cast<const char*>(Kimg + (pk & 1) * NP * HD);
#pragma unroll
                    for (int dd = 0; dd < 2; ++dd) {
                        const char* kp0 = Kc + (widx ? kboff[2 * dd + 1][0] : kboff[2 * dd][0]);
                        const char* kp1 = Kc + (widx ? kboff[2 * dd + 1][1] : kboff[2 * dd][1]);
#pragma unroll
                        for (int u = 0; u < NTP; ++u)
                            kt[dd][u] = lds_tr_frag_at(reinterpret_cast<const bf16_t*>(kp0 + u * 32 * HD * 2), reinterpret_cast<const bf16_t*>(kp1 + u * 32 * HD * 2));
                    }
                }
#pragma unroll
                for (int qs = 0; qs < 2; ++qs) {
                    if (qb0 + 16 * qs < N) {   // wave-uniform
                        // B[k = key][col = query] = dS^T, shared by the wave's two d tiles; the two MFMA chains are independent
                        bf16x8_t sb[NTP];
                        const char* sp0 = dSb + sboff[qs][0];
                        const char* sp1 = dSb + sboff[qs][1];
#pragma unroll
                        for (int u = 0; u < NTP; ++u)
                            sb[u] = lds_tr_frag_at(reinterpret_cast<const bf16_t*>(sp0 + u * 32 * 64), reinterpret_cast<const bf16_t*>(sp1 + u * 32 * 64));
                        float4_t dq0 = (float4_t){0.f, 0.f, 0.f, 0.f}, dq1 = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int u = 0; u < NTP; ++u) {
                            dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt[0][u], sb[u], dq0, 0, 0, 0);
                            dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt[1][u], sb[u], dq1, 0, 0, 0);
                        }
                        // D[row = d 4g+r][col = query i]
                        const int q = qb0 + 16 * qs + i;
                        if (q < N) {
                            bf16_t* dst = dqkv + ((int64_t)b * N + q) * D3 + h * HD + 16 * widx + 4 * g;
                            uint2 w;
                            w.x = pack_bf16x2(dq0[0] * scale, dq0[1] * scale);
                            w.y = pack_bf16x2(dq0[2] * scale, dq0[3] * scale);
                            *reinterpret_cast<uint2*>(dst) = w;
                            w.x = pack_bf16x2(dq1[0] * scale, dq1[1] * scale);
                            w.y = pack_bf16x2(dq1[2] * scale, dq1[3] * scale);
                            *reinterpret_cast<uint2*>(dst + 32) = w;
                        }
                    }
                }
            }
            if (dbg_period <= 1 || p % dbg_period == 0) lds_barrier();   // dbg_period > 1: timing experiment only (results WRONG)
            if (bh < 0) break;
            pk = k;
            pj = j;
            pbh = bh;
            if (++j == NQB) {
                j = 0;
                ++k;
            }
        }
    } else {
        // ================================================================ producer
        // Iteration p, between the barriers that end steps p - 1 and p:
        //   1. wait until at most the NDMA pieces issued in iteration p - 1 are outstanding (a wave's loads, stores and LDS-DMA retire
        //      in issue order): everything issued in iteration p - 2 or earlier has landed - the ring pieces of step p + 1;
        //   2. delta / scaled lse of step p + 1 from the landed pieces (published, like the pieces, by the barrier that ends step p);
        //   3. issue the ring pieces of step p + 3 (into the slots step p - 1 has just released).
        int k = 0, j = 0, bh = head0;
        for (int p = 0;; ++p) {
            if (j == 0 && p) bh = heads[k & 3];
            if (bh >= 0 && !dbg_no_prod) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
                if (j < NQB - 1 || heads[(k + 1) & 3] >= 0) prod_aux(p + 1, p + 1);       // step p + 1 exists
                prod_dma(p + 3, p + 3);
            }
            if (dbg_period <= 1 || p % dbg_period == 0) lds_barrier();   // dbg_period > 1: timing experiment only (results WRONG)
            if (bh < 0) break;
            if (++j == NQB) {
                j = 0;
                ++k;
            }
        }
    }
}

// ------------------------------------------------------------------------------------ forward, persistent pipelined kernel
// 193 <= N <= 208, the forward counterpart of attn_bwd_pipe_kernel: one 16-wave workgroup per CU walks heads drawn from a device
// counter.  Waves 0 .. 12 own one 16-query tile each and run exactly the chunk body of attn_fwd_head_kernel (same arithmetic, same
// order: outputs, lse and keep bits are bit-identical); waves 13 / 14 / 15 bring the NEXT head's K / V / Q rows into the other half of
// double-buffered LDS images by LDS-DMA (inline asm pieces, see glds16_pipe_s) while the current head is computed - the compute waves
// issue no load at all (their Q fragments come from the Q image), only fire-and-forget stores, and meet the loaders at one LDS
// barrier per head.  What the whole-head kernel leaves idle: its two 7-wave workgroups per CU each wait for their own K / V, then
// for their own Q rows, before the first MFMA, and a wave carries two query tiles (14 slots for 13 tiles).
namespace fpipe {
constexpr int NKT = 13;
constexpr int IMG = NKT * 16 * HD;                    // bf16 elements of one image (208 rows)
constexpr size_t OFF_K = 0, OFF_V = OFF_K + (size_t)2 * IMG * 2, OFF_Q = OFF_V + (size_t)2 * IMG * 2, OFF_HEADS = OFF_Q + (size_t)2 * IMG * 2;
constexpr size_t LDS_BYTES = OFF_HEADS + 16;          // 159,760 bytes
}  // namespace fpipe
__device__ unsigned int g_fpipe_head_ctr[32];

template <bool DROP>
__global__ void __launch_bounds__(1024, 4) attn_fwd_pipe_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse_out,
                                                             uint32_t* __restrict__ bits_out, unsigned int* __restrict__ head_ctr, int N, int H, int total_heads,
                                                             int Np4, float scale_log2, float drop_scale, uint32_t drop_thr, uint32_t drop_key) {
    using namespace fpipe;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* Kimg = reinterpret_cast<bf16_t*>(smem_raw + OFF_K);
    bf16_t* Vimg = reinterpret_cast<bf16_t*>(smem_raw + OFF_V);
    bf16_t* Qimg = reinterpret_cast<bf16_t*>(smem_raw + OFF_Q);
    int* heads = reinterpret_cast<int*>(smem_raw + OFF_HEADS);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int Dm = H * HD;
    const int64_t D3 = 3 * (int64_t)Dm;
    const int r8 = lane >> 3, c8 = lane & 7;

    auto claim = [&]() -> int {            // wave 13, uniform: one ticket = one head index (see attn_bwd_pipe_kernel)
        unsigned t = 0;
        if (lane == 0) {
            t = atomicAdd(head_ctr, 1u);
            if (t == (unsigned)total_heads + 2u * gridDim.x - 1u) atomicExch(head_ctr, 0u);
        }
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        return t < (unsigned)total_heads ? (int)t : -1;
    };
    // the 26 pieces (8 rows x 128 bytes) of one image of head bh: which = 0 Q (swz_row), 1 K (swz_row), 2 V (swz_trv)
    auto load_image = [&](int bh, int which, bf16_t* img) {
        const int b = bh / H, h = bh - b * H;
        const bf16_t* sbase = qkv + (int64_t)b * N * D3 + (int64_t)which * Dm + h * HD;
#pragma unroll 1
        for (int inst = 0; inst < NKT * 2; ++inst) {
            const int r = inst * 8 + r8;
            const int sw = which == 2 ? swz_trv(r) : swz_row(r);
            const uint32_t voff = (__umul24((uint32_t)min(r, N - 1), (uint32_t)(3 * Dm)) + (uint32_t)((c8 ^ sw) << 3)) * 2u;
            glds16_pipe_s(sbase, voff, img + inst * 512);
        }
    };
    if (wave == 13) {
        const int h0 = claim();
        const int h1 = claim();
        if (lane == 0) {
            heads[0] = h0;
            heads[1] = h1;
            heads[2] = -1;
            heads[3] = -1;
        }
    }
    __syncthreads();
    const int head0 = heads[0];
    if (head0 < 0) return;
    if (wave == 13) load_image(head0, 1, Kimg);
    else if (wave == 14) load_image(head0, 2, Vimg);
    else if (wave == 15) load_image(head0, 0, Qimg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (wave < NKT) {
        // ================================================================ compute: query tile `wave` of every head
        const int nkt = (N + 15) >> 4;
        const int q = 16 * wave + i, qc = min(q, N - 1);
        int bh = head0;
        for (int k = 0;; ++k) {
            if (k) bh = heads[k & 3];
            if (bh >= 0) {
                const bf16_t* Ks = Kimg + (k & 1) * IMG;
                const bf16_t* Vs = Vimg + (k & 1) * IMG;
                const bf16_t* Qs = Qimg + (k & 1) * IMG;
                const bf16x8_t qf0 = lds_row_frag(Qs, q, g);
                const bf16x8_t qf1 = lds_row_frag(Qs, q, 4 + g);
                const uint32_t cbase = ((((uint32_t)bh * (uint32_t)N + (uint32_t)qc) * (uint32_t)Np4) >> 1) + 2u * (uint32_t)g;
                float m_run = -INFINITY, l_run = 0.f;
                float4_t oacc[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oacc[dt] = (float4_t){0.f, 0.f, 0.f, 0.f};
                uint32_t bits_lo = 0u, bits_hi = 0u;
                const int nfull = N >> 6;
                const int nchunk = (nkt + 3) >> 2;
                FwdState st{m_run, l_run, oacc, bits_lo, bits_hi};
#pragma unroll 1
                for (int kc = 0; kc < nfull; ++kc)
                    fwd_chunk<true, DROP>(st, Ks, Vs, qf0, qf1, kc, 4, N, g, i, cbase, scale_log2, drop_thr, drop_key);
                if (nfull < nchunk) fwd_chunk<false, DROP>(st, Ks, Vs, qf0, qf1, nfull, nkt - 4 * nfull, N, g, i, cbase, scale_log2, drop_thr, drop_key);
                float sum = l_run;
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float oscale = (DROP ? drop_scale : 1.0f) / sum;
                if (q < N) {
                    const int b = bh / H, h = bh - b * H;
                    if (g == 0) lse_out[(int64_t)bh * N + q] = (m_run * scale_log2 + log2f(sum)) * 0.69314718055994530942f;
                    bf16_t* op = o + ((int64_t)b * N + q) * Dm + h * HD;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        uint2 w;
                        w.x = pack_bf16x2(oacc[dt][0] * oscale, oacc[dt][1] * oscale);
                        w.y = pack_bf16x2(oacc[dt][2] * oscale, oacc[dt][3] * oscale);
                        *reinterpret_cast<uint2*>(op + 16 * dt + 4 * g) = w;
                    }
                    if (DROP && bits_out) *reinterpret_cast<uint2*>(bits_out + (((int64_t)bh * N + q) * 4 + g) * 2) = make_uint2(bits_lo, bits_hi);
                }
            }
            lds_barrier();
            if (bh < 0) break;
        }
    } else {
        // ================================================================ loaders: the next head's K (13) / V (14) / Q (15) image
        int bh = head0;
        for (int k = 0;; ++k) {
            if (k) bh = heads[k & 3];
            if (bh >= 0) {
                if (wave == 13) {                      // this workgroup's head after next
                    const int h2 = claim();
                    if (lane == 0) heads[(k + 2) & 3] = h2;
                }
                const int nxt = heads[(k + 1) & 3];
                if (nxt >= 0) {
                    const int half = (k + 1) & 1;
                    if (wave == 13) load_image(nxt, 1, Kimg + half * IMG);
                    else if (wave == 14) load_image(nxt, 2, Vimg + half * IMG);
                    else load_image(nxt, 0, Qimg + half * IMG);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // landed before the barrier that hands the images over
                }
            }
            lds_barrier();
            if (bh < 0) break;
        }
    }
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
    const uint32_t ebase = ((uint32_t)bh * (uint32_t)N + (uint32_t)qc) * (uint32_t)((N + 3) & ~3);

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
                    keep4_even(ebase + (uint32_t)(k0 + 16 * t + 4 * g), drop_key, drop_thr, 1.0f, keepc);   // ebase, key offset: multiples of 4
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
        bf16_t* dst = (role_g ? Gb[buf] : Qb[buf]) + sr * HD + ((sc ^ swz_dual(sr)) << 3);
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
                    sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag_dual(Qs, 16 * qs + i, 4 * ks + g), kfr[ks], sv, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag_dual(Gs, 16 * qs + i, 4 * ks + g), vfr[ks], dp, 0, 0, 0);
                }
                // keys i and i^1 of one query share a hash word (the row stride is even): the even lane hashes queries r = 0, 1, the
                // odd lane r = 2, 3, and a quad_perm [1,0,3,2] hands each the other's two
                uint32_t hw[4];
                if (DROP) {
                    uint32_t hm[2], ho[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int ql = 16 * qs + 4 * g + 2 * (i & 1) + j;
                        const uint32_t e = ((uint32_t)bh * (uint32_t)N + (uint32_t)min(q0 + ql, N - 1)) * (uint32_t)((N + 3) & ~3) + (uint32_t)min(key, N - 1);
                        hm[j] = chb_hash32((e >> 1) ^ drop_key);
                        ho[j] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hm[j], 0xB1, 0xF, 0xF, true);
                    }
                    const bool odd_lane = (i & 1) != 0;
                    hw[0] = odd_lane ? ho[0] : hm[0];
                    hw[1] = odd_lane ? ho[1] : hm[1];
                    hw[2] = odd_lane ? hm[0] : ho[0];
                    hw[3] = odd_lane ? hm[1] : ho[1];
                }
                const bool key_odd = (min(key, N - 1) & 1) != 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = 16 * qs + 4 * g + r;
                    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], scale_log2, -l2b[cur][ql]));   // pad queries: lse = +inf -> 0
                    if (!tile_full) p = (key < N) ? p : 0.f;
                    float keepc = 1.0f;
                    if (DROP) {
                        const uint32_t u = key_odd ? (hw[r] >> 16) : (hw[r] & 0xffffu);
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
                const bf16x8_t gt = lds_tr_frag_dual(Gs, 4 * g, 16 + 4 * g, 16 * dt, i);
                const bf16x8_t qt = lds_tr_frag_dual(Qs, 4 * g, 16 + 4 * g, 16 * dt, i);
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
    const uint32_t ebase = ((uint32_t)bh * (uint32_t)N + (uint32_t)qc) * (uint32_t)((N + 3) & ~3);

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
        const int off = sr * HD + ((sc ^ swz_dual(sr)) << 3);
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
                            sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag_dual(Ks, 16 * t + i, 4 * ks + g), qb[ks], sv, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag_dual(Vs, 16 * t + i, 4 * ks + g), gb[ks], dp, 0, 0, 0);
                        }
                        float keepc[4] = {1.f, 1.f, 1.f, 1.f};
                        if (DROP) keep4_even(ebase + (uint32_t)key0, drop_key, drop_thr, drop_scale, keepc);   // ebase, key0: multiples of 4
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
                        const bf16x8_t kt = lds_tr_frag_dual(Ks, 32 * u + 4 * g, 32 * u + 16 + 4 * g, 16 * dt, i);
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


// Head-counter slot of a persistent attention launch: one slot per (device, stream) - launches of ONE stream run one after the
// other and each leaves its slot at zero, so they may share it; launches of different streams may overlap and never share one.  -1
// when all slots are taken by other streams (the caller then runs the non-persistent kernel).
static int pipe_ctr_slot(int dev, hipStream_t s) {
    static std::mutex mu;
    static struct { int dev; hipStream_t s; bool used; } table[32];
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < 32; ++i)
        if (table[i].used && table[i].dev == dev && table[i].s == s) return i;
    for (int i = 0; i < 32; ++i)
        if (!table[i].used) {
            table[i].used = true;
            table[i].dev = dev;
            table[i].s = s;
            return i;
        }
    return -1;
}

}  // namespace

extern "C" {

int chb_attention_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int hd, float drop_rate, uint32_t drop_key,
                      uint32_t* drop_bits, void* stream) {
    if (!qkv || !o || !lse || B < 0 || N <= 0 || H <= 0 || drop_rate < 0.f || drop_rate >= 1.f) return CHB_EINVAL;
    if (hd != HD) return CHB_EUNSUPPORTED;
    if ((double)B * H * N * ((N + 3) & ~3) >= 4294967296.0) return CHB_EUNSUPPORTED;   // dropout element index is 32-bit
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
    // default for N <= 224: the whole-head kernel (K / V in LDS once, online softmax, two hashes per four keys, keep bits saved for
    // the backward).  CHB_ATTN_FWD_ALGO = 1 | 2 forces the resident (N <= 224) | streaming kernel: the parity tests cross-check.
    // Keep bits asked for (drop_bits with dropout on): only the whole-head kernel writes them, so it runs whatever the A/B switch
    // says - a backward that tested bits nobody wrote would silently use another mask than the forward.  Beyond 224 tokens no
    // kernel writes them: refused.
    const bool want_bits = drop_bits != nullptr && thr != 0u;
    if (want_bits && N > 224) return CHB_EUNSUPPORTED;
    // 193 <= N <= 208 (ViT at 224^2): the persistent pipelined kernel, one workgroup per CU (CHB_ATTN_FWD_ALGO = 3 keeps the
    // whole-head kernel for A/B; 1 / 2 still force the resident / streaming kernels when no keep bits are asked for)
    if (N >= 193 && N <= 16 * fpipe::NKT && (algo == 0 || algo == 4 || ((algo == 1 || algo == 2) && want_bits))) {
        static std::atomic<int> n_cus[64];
        static std::atomic<unsigned int*> ctr_of[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return CHB_ELAUNCH;
        int cus = n_cus[dev].load(std::memory_order_acquire);
        if (cus == 0) {
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return CHB_ELAUNCH;
            if (hipFuncSetAttribute((const void*)attn_fwd_pipe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fpipe::LDS_BYTES) != hipSuccess ||
                hipFuncSetAttribute((const void*)attn_fwd_pipe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fpipe::LDS_BYTES) != hipSuccess)
                return CHB_ELAUNCH;
            unsigned int* base = nullptr;
            if (hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_fpipe_head_ctr)) != hipSuccess || !base) return CHB_ELAUNCH;
            ctr_of[dev].store(base, std::memory_order_release);
            n_cus[dev].store(cus, std::memory_order_release);
        }
        const int slot = pipe_ctr_slot(dev, s);
        unsigned int* ctr = ctr_of[dev].load(std::memory_order_acquire) + (slot < 0 ? 0 : slot);
        const int total = B * H;
        const dim3 pgrid(total < cus ? total : cus);
        const int np4 = (N + 3) & ~3;
        if (slot >= 0) {        // (none left: more than 32 streams run attention on this device - the whole-head kernel below)
            if (thr) hipLaunchKernelGGL((attn_fwd_pipe_kernel<true>), pgrid, dim3(1024), fpipe::LDS_BYTES, s, in, out, lse, drop_bits, ctr, N, H, total, np4, scale_log2, ds, thr, drop_key);
            else hipLaunchKernelGGL((attn_fwd_pipe_kernel<false>), pgrid, dim3(1024), fpipe::LDS_BYTES, s, in, out, lse, drop_bits, ctr, N, H, total, np4, scale_log2, ds, thr, drop_key);
            CHB_LAUNCH_CHECK();
            return CHB_OK;
        }
    }
    if (N <= 224 && (want_bits || (algo != 1 && algo != 2))) {     // whole-head kernel (also algo 3: the A side of the pipelined kernel)
        const int nkt = (N + 15) >> 4;
        const int nw = (nkt + 1) >> 1;
        const size_t lds = (size_t)2 * nkt * 16 * HD * sizeof(bf16_t);
        static std::atomic<bool> attr_set{false};
        if (!attr_set.load(std::memory_order_acquire)) {
            if (hipFuncSetAttribute((const void*)attn_fwd_head_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
                hipFuncSetAttribute((const void*)attn_fwd_head_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess)
                return CHB_ELAUNCH;
            attr_set.store(true, std::memory_order_release);
        }
        const int np4 = (N + 3) & ~3;
        if (thr) hipLaunchKernelGGL((attn_fwd_head_kernel<true>), grid, dim3(64 * nw), lds, s, in, out, lse, drop_bits, N, H, np4, scale_log2, ds, thr, drop_key);
        else hipLaunchKernelGGL((attn_fwd_head_kernel<false>), grid, dim3(64 * nw), lds, s, in, out, lse, drop_bits, N, H, np4, scale_log2, ds, thr, drop_key);
        CHB_LAUNCH_CHECK();
        return CHB_OK;
    }
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
                      float drop_rate, uint32_t drop_key, float* dbias_qkv, float* dbias_ws, const uint32_t* drop_bits, void* stream) {
    if (!qkv || !o || !d_o || !lse || !dqkv || B < 0 || N <= 0 || H <= 0 || drop_rate < 0.f || drop_rate >= 1.f) return CHB_EINVAL;
    if (hd != HD) return CHB_EUNSUPPORTED;
    if ((double)B * H * N * ((N + 3) & ~3) >= 4294967296.0) return CHB_EUNSUPPORTED;
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
    // 193 <= N <= 208 (ViT at 224^2: 197 / 198 tokens), no fused bias gradient, dropout through the forward's keep bits or none: the
    // persistent pipelined kernel, one workgroup per CU (CHB_ATTN_BWD_ALGO = 4 keeps the lean one-workgroup-per-head kernel for A/B)
    if (!dbias_qkv && (bwd_algo == 0 || bwd_algo == 5) && N >= 193 && N <= 16 * pipe::NKT && (!thr || drop_bits)) {
        static std::atomic<int> n_cus[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return CHB_ELAUNCH;
        int cus = n_cus[dev].load(std::memory_order_acquire);
        if (cus == 0) {
            // first launch on this device: CU count, and the kernels' dynamic LDS limit (both per device)
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return CHB_ELAUNCH;
            if (hipFuncSetAttribute((const void*)attn_bwd_pipe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pipe::LDS_BYTES) != hipSuccess ||
                hipFuncSetAttribute((const void*)attn_bwd_pipe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pipe::LDS_BYTES) != hipSuccess)
                return CHB_ELAUNCH;
            n_cus[dev].store(cus, std::memory_order_release);
        }
        const int total = B * H;
        const dim3 pgrid(total < cus ? total : cus);
        static std::atomic<unsigned int*> ctr_of[64];           // device address of g_pipe_head_ctr, looked up once per device
        unsigned int* ctr_base = ctr_of[dev].load(std::memory_order_acquire);
        if (!ctr_base) {
            if (hipGetSymbolAddress((void**)&ctr_base, HIP_SYMBOL(g_pipe_head_ctr)) != hipSuccess || !ctr_base) return CHB_ELAUNCH;
            ctr_of[dev].store(ctr_base, std::memory_order_release);
        }
        const int slot = pipe_ctr_slot(dev, s);
        if (slot >= 0) {        // (none left: more than 32 streams run attention on this device - the lean kernel below)
            unsigned int* ctr = ctr_base + slot;
            if (thr) hipLaunchKernelGGL((attn_bwd_pipe_kernel<true>), pgrid, dim3(1024), pipe::LDS_BYTES, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)d_o, lse,
                                        (bf16_t*)dqkv, drop_bits, ctr, N, H, total, scale, scale_log2, ds, chb_option(CHB_OPT_DEBUG));
            else hipLaunchKernelGGL((attn_bwd_pipe_kernel<false>), pgrid, dim3(1024), pipe::LDS_BYTES, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)d_o, lse,
                                    (bf16_t*)dqkv, drop_bits, ctr, N, H, total, scale, scale_log2, ds, chb_option(CHB_OPT_DEBUG));
            CHB_LAUNCH_CHECK();
            return CHB_OK;
        }
    }
    // otherwise (no fused bias gradient): the lean kernel; it tests the forward's keep bits when they are given
    if (!dbias_qkv && (bwd_algo == 0 || bwd_algo == 4 || bwd_algo == 5)) {
#define CHB_BWDH_V(NTP, DROP, BITS)                                                                                              \
    do {                                                                                                                         \
        const size_t lds = bwd_head_lds_bytes<NTP>();                                                                            \
        static std::atomic<bool> attr_set{false};                                                                                \
        if (!attr_set.load(std::memory_order_acquire)) {                                                                         \
            if (hipFuncSetAttribute((const void*)attn_bwd_head_kernel<NTP, DROP, BITS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
                return CHB_ELAUNCH;                                                                                              \
            attr_set.store(true, std::memory_order_release);                                                                     \
        }                                                                                                                        \
        hipLaunchKernelGGL((attn_bwd_head_kernel<NTP, DROP, BITS>), grid, dim3(1024), lds, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)d_o, \
                           lse, (bf16_t*)dqkv, drop_bits, N, H, scale, scale_log2, ds, thr, drop_key, chb_option(CHB_OPT_DEBUG));     \
    } while (0)
#define CHB_BWDH(NTP)                                          \
    do {                                                       \
        if (!thr) CHB_BWDH_V(NTP, false, false);               \
        else if (drop_bits) CHB_BWDH_V(NTP, true, true);       \
        else CHB_BWDH_V(NTP, true, false);                     \
    } while (0)
        switch ((N + 31) >> 5) {
            case 1: CHB_BWDH(1); break;
            case 2: CHB_BWDH(2); break;
            case 3: CHB_BWDH(3); break;
            case 4: CHB_BWDH(4); break;
            case 5: CHB_BWDH(5); break;
            case 6: CHB_BWDH(6); break;
            default: CHB_BWDH(7); break;
        }
#undef CHB_BWDH
#undef CHB_BWDH_V
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
