// bf16 MFMA GEMMs for the ViT block on gfx950 (CDNA4): v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//
//  chb_gemm_nt : C[M,N] = epi(A[M,K] . B[N,K]^T)   forward (B = W^T copy) and dgrad (B = W)
//  chb_gemm_tn : dW[Kd,Nd] += X[M,Kd]^T . dY[M,Nd]  wgrad, reduction over the long token axis
//
// Both: 128x128 output tile per 256-thread workgroup (4 waves, 64x64 per wave, 4x4 MFMA
// tiles), 64-deep reduction steps, operands staged global->LDS with global_load_lds_dwordx4
// (no VGPR round trip) into a double buffer.  LDS images are lane-linear (the DMA writes
// base + lane*16), so bank conflicts are removed by permuting the per-lane SOURCE chunk and
// applying the same XOR on the fragment reads.  Operands are swapped in the MFMA so each
// lane ends up with 4 consecutive output columns (8/16-byte stores).  Workgroup ids are
// remapped so the workgroups that share an A row-panel run on one XCD (one L2).
//
// Replaces: tf.einsum / Dense / Conv2D op sequences of layers/attention.py:113-125,
// layers/transformer.py:72-77, models/backbones/vision_transformer.py:235-283.
#include "common.hpp"
#include "../../include/chambers_hip.h"
#include <stdlib.h>
#include <atomic>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
// tile rows of epilogue inputs (residual / saved gelu') requested ahead of the row being written; -D overrides are for A/B builds
#ifndef EPI_DEPTH_RESID
#define EPI_DEPTH_RESID 1
#endif
#ifndef EPI_DEPTH_DGELU
#define EPI_DEPTH_DGELU 1
#endif
#ifndef EPI_DEPTH_DGELU_WIDE
#define EPI_DEPTH_DGELU_WIDE 0     // the 8-column form: one row ahead costs 8 more registers per lane and spills
#endif
constexpr int TILE_ELEMS = 128 * 64;  // one operand tile, either orientation

struct GemmParams {
    const bf16_t* A; int64_t lda;
    const bf16_t* B; int64_t ldb;
    void* C; int64_t ldc;
    int M, N, K;
    const float* bias;
    bf16_t* aux; int64_t ld_aux;
    const float* resid; int64_t ld_resid;
    int period;      // PATCH: patches per image
    int n_special;   // PATCH: tokens in front of the patches (1 = class token, 2 = class + distillation token)
    float drop_scale; uint32_t drop_thr; uint32_t drop_key;
    int tiles_m, tiles_n;
    float* colsum;   // optional fp32 [N]: += column sums of the output (bias gradient of the consumer layer)
    int walk_panel;  // persistent 256x256 kernel: XCD-panel tile walk (see TileWalk)
    int queue_slot;  // persistent 256x256 kernel: >= 0 = tiles after a workgroup's first are claimed from per-XCD counters of this slot
};

// bijective XCD-aware remap: consecutive virtual ids (which share an A panel) stay on one XCD
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int x = bid & 7, idx = bid >> 3;
    const int start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return start + idx;
}

#ifdef CHB_CLOCK_STAMPS
// Diagnostic build only (tools/gemm_clock.py through tools/ab_build.sh -DCHB_CLOCK_STAMPS): wave 0 of every workgroup stamps the
// shader clock counter and the 100 MHz reference counter at kernel entry and exit into a buffer nothing else reads - the clock the
// chip held over the launch is d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back 6).
__device__ unsigned long long g_clock_stamps[1024][4];
__device__ __forceinline__ void clock_stamp(int which) {
    if (threadIdx.x == 0) {
        g_clock_stamps[blockIdx.x & 1023][which] = __builtin_amdgcn_s_memtime();
        g_clock_stamps[blockIdx.x & 1023][which + 1] = __builtin_amdgcn_s_memrealtime();
    }
}
#define CLOCK_STAMP(w) clock_stamp(w)
#else
#define CLOCK_STAMP(w)
#endif
#ifdef CHB_PHASE_STAMPS
// Second diagnostic build (tools/gemm_phases.py): every wave reads the cycle counter at seven points of gemm_nt256_kernel's K-step
// and sums the segments; waves 0 and 5 of each workgroup leave their sums in a buffer of their own.  Costs ~10 % of the loop.
__device__ unsigned int g_phase_stamps[1024][2][10];
#define PHASE_DECL unsigned int ph_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ph_prev = __builtin_amdgcn_s_memtime();
#define PHASE(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_sum[i] += (unsigned int)(t_ - ph_prev); ph_prev = t_; }
#define PHASE_FLUSH if (lane == 0 && (wave == 0 || wave == 5)) { for (int q_ = 0; q_ < 10; ++q_) g_phase_stamps[blockIdx.x & 1023][wave == 5][q_] = ph_sum[q_]; }
#else
#define PHASE_DECL
#define PHASE(i)
#define PHASE_FLUSH
#endif

__device__ __forceinline__ void glds16(const bf16_t* src, bf16_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_wave_base), 16, 0, 0);
}

// ---- NT: both operand tiles are [128 rows][64 k] (128-byte rows, 8 chunks of 16 B) ----------
// LDS position (row r, chunk c') holds global chunk c' ^ ((r >> 1) & 7).
__device__ __forceinline__ void stage_nt(const bf16_t* __restrict__ g, int64_t ld, int row0, int max_row, int k0,
                                         bf16_t* lds_tile, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int inst = j * 4 + wave;           // 16 instructions of 8 rows each
        const int r = inst * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < max_row ? gr : max_row - 1;    // edge rows: re-read a valid row, masked at store
        glds16(g + (int64_t)gr * ld + k0 + c * 8, lds_tile + inst * 512);
    }
}

__device__ __forceinline__ bf16x8_t frag_nt(const bf16_t* lds_tile, int r, int chunk) {
    const int off = r * 64 + ((chunk ^ ((r >> 1) & 7)) << 3);
    return *reinterpret_cast<const bf16x8_t*>(lds_tile + off);
}

// Epilogue of one 16-row MFMA tile row: this lane owns `row` and, for each of NB column tiles,
// 4 consecutive columns col0 + 16*b .. +3.  All global loads of the row (residual / saved
// pre-activation) are issued together before any store so they overlap instead of serialising.
template <int EPI, int OUT, int NB, bool GUARD>
__device__ __forceinline__ void epilogue_row(const GemmParams& p, int row, int col0, const float4_t* acc, const float4* bias4) {
    float v[NB][4];
    float4 r4[NB];
    uint2 a2[NB];
    int64_t orow = row;
    const float* posrow = nullptr;
    if (EPI == CHB_EPI_PATCH) {
        const int b = row / p.period, pp = row - b * p.period;
        orow = (int64_t)b * (p.period + p.n_special) + p.n_special + pp;
        posrow = p.resid + (int64_t)(p.n_special + pp) * p.ld_resid;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int col = col0 + 16 * b;
        if (!GUARD || col < p.N) {
            if (EPI == CHB_EPI_RESID) r4[b] = *reinterpret_cast<const float4*>(p.resid + (int64_t)row * p.ld_resid + col);
            if (EPI == CHB_EPI_PATCH) r4[b] = *reinterpret_cast<const float4*>(posrow + col);
            if (EPI == CHB_EPI_DGELU) a2[b] = *reinterpret_cast<const uint2*>(p.aux + (int64_t)row * p.ld_aux + col);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int col = col0 + 16 * b;
        if (GUARD && col >= p.N) continue;
        v[b][0] = acc[b][0] + bias4[b].x; v[b][1] = acc[b][1] + bias4[b].y;
        v[b][2] = acc[b][2] + bias4[b].z; v[b][3] = acc[b][3] + bias4[b].w;
        if (EPI == CHB_EPI_GELU) {
            float d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) gelu_both(v[b][i], v[b][i], d[i]);
            uint2 a;
            a.x = pack_bf16x2(d[0], d[1]);
            a.y = pack_bf16x2(d[2], d[3]);
            *reinterpret_cast<uint2*>(p.aux + (int64_t)row * p.ld_aux + col) = a;
        } else if (EPI == CHB_EPI_DGELU) {
            v[b][0] *= bf16_to_f32((bf16_t)(a2[b].x & 0xffff));
            v[b][1] *= bf16_to_f32((bf16_t)(a2[b].x >> 16));
            v[b][2] *= bf16_to_f32((bf16_t)(a2[b].y & 0xffff));
            v[b][3] *= bf16_to_f32((bf16_t)(a2[b].y >> 16));
        } else if (EPI == CHB_EPI_RESID || EPI == CHB_EPI_PATCH) {
            if (EPI == CHB_EPI_PATCH) {  // + positional embedding, then dropout
                v[b][0] += r4[b].x; v[b][1] += r4[b].y; v[b][2] += r4[b].z; v[b][3] += r4[b].w;
            }
            if (p.drop_thr) {
                const uint64_t e0 = (uint64_t)orow * (uint64_t)p.N + (uint64_t)col;  // even (N % 4 == 0)
                bool k0, k1, k2, k3;
                chb_keep2((uint32_t)(e0 >> 1), p.drop_key, p.drop_thr, k0, k1);
                chb_keep2((uint32_t)(e0 >> 1) + 1u, p.drop_key, p.drop_thr, k2, k3);
                v[b][0] = k0 ? v[b][0] * p.drop_scale : 0.0f;
                v[b][1] = k1 ? v[b][1] * p.drop_scale : 0.0f;
                v[b][2] = k2 ? v[b][2] * p.drop_scale : 0.0f;
                v[b][3] = k3 ? v[b][3] * p.drop_scale : 0.0f;
            }
            if (EPI == CHB_EPI_RESID) {
                v[b][0] += r4[b].x; v[b][1] += r4[b].y; v[b][2] += r4[b].z; v[b][3] += r4[b].w;
            }
        }
        if (OUT == CHB_OUT_F32) {
            *reinterpret_cast<float4*>((float*)p.C + orow * p.ldc + col) = make_float4(v[b][0], v[b][1], v[b][2], v[b][3]);
        } else {
            uint2 o;
            o.x = pack_bf16x2(v[b][0], v[b][1]);
            o.y = pack_bf16x2(v[b][2], v[b][3]);
            *reinterpret_cast<uint2*>((bf16_t*)p.C + orow * p.ldc + col) = o;
        }
    }
}

template <int NB>
__device__ __forceinline__ void load_bias(const GemmParams& p, int col0, float4* bias4) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int col = col0 + 16 * b;
        bias4[b] = (p.bias && col < p.N) ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int EPI, int OUT>
__global__ void __launch_bounds__(256) gemm_nt_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * TILE_ELEMS];  // A0 B0 A1 B1 (64 KiB)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int v = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tm = v / p.tiles_n, tn = v - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, i = lane & 15;

    float4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int nt = p.K / BK;
    stage_nt(p.A, p.lda, m0, p.M, 0, smem, wave, lane);
    stage_nt(p.B, p.ldb, n0, p.N, 0, smem + TILE_ELEMS, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nt; ++kt) {
        bf16_t* As = smem + (kt & 1) * 2 * TILE_ELEMS;
        bf16_t* Bs = As + TILE_ELEMS;
        if (kt + 1 < nt) {
            bf16_t* An = smem + ((kt + 1) & 1) * 2 * TILE_ELEMS;
            stage_nt(p.A, p.lda, m0, p.M, (kt + 1) * BK, An, wave, lane);
            stage_nt(p.B, p.ldb, n0, p.N, (kt + 1) * BK, An + TILE_ELEMS, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[4], bfr[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = frag_nt(As, wm * 64 + a * 16 + i, ks * 4 + g);
#pragma unroll
            for (int b = 0; b < 4; ++b) bfr[b] = frag_nt(Bs, wn * 64 + b * 16 + i, ks * 4 + g);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    // swapped operands: D[row = n][col = m] -> lane holds 4 consecutive n for one m
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    float4 bias4[4];
    load_bias<4>(p, n0 + wn * 64 + g * 4, bias4);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int row = m0 + wm * 64 + a * 16 + i;
        if (row < p.M) epilogue_row<EPI, OUT, 4, true>(p, row, n0 + wn * 64 + g * 4, acc[a], bias4);
    }
}

// ---- NT, persistent 256x256 tiles -------------------------------------------------------------
// One 512-thread workgroup per CU (8 waves = 2(M) x 4(N), 128x64 outputs per wave) walks a list
// of 256x256 output tiles.  Each 64-deep K-tile is four 16 KiB half-tiles (A rows 0-127 / 128-255,
// B rows 0-127 / 128-255) in a 2-deep LDS ring (128 KiB).  A K-tile is computed in two phases of
// 32 MFMAs (rows 0-63 / 64-127 of the wave's output) with ONE barrier per K-tile; the loads of later
// K-tiles stay in flight across it and across output-tile boundaries (the epilogue of tile i overlaps
// the loads of tile i+1):
//   phase 1: stage A of step s+1 into the other ring | read B (all 64 columns) and A rows 0-63 | MFMAs
//   phase 2: read A rows 64-127 | vmcnt(0) + barrier | stage B of step s+2 into THIS ring | MFMAs
// The barrier follows every ds_read of this ring (each wave waits lgkmcnt(0) before its MFMAs), so the B
// slots can be restaged right behind it and the A slots in phase 1 of the next step; the wait in front
// of it retires step s+1 (A issued one phase earlier, B one step earlier), first read after the MFMAs.
// Epilogue of the persistent kernel: the wave's 128x64 fp32 accumulators go through a wave-private
// 4 KiB LDS scratch, one 16x64 MFMA tile row at a time, and come back row-major (lane = 4 rows x 16
// chunks of 4 columns), so that EVERY global access of the epilogue (residual / saved pre-activation
// loads, output and aux stores) is 4 rows x 128..256 contiguous bytes per wave-instruction instead of
// 16 rows x 32 bytes.  The scratch is XOR-swizzled (16-byte chunk ^ row) - conflict-free both ways.
template <int EPI, int OUT, bool GUARD, int A0, int A1>
__device__ __forceinline__ void epilogue_staged(const GemmParams& p, float* stage, int m_base, int n_base, float4_t (&acc)[8][4],
                                                int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int cr = lane >> 4, c4 = lane & 15;
    const int col = n_base + c4 * 4;
    const bool colok = !GUARD || col < p.N;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && colok) bias = *reinterpret_cast<const float4*>(p.bias + col);
    // lane-constant pieces of every address (the per-(a,k) part is a compile-time row count times a uniform stride)
    constexpr int ESZ = (OUT == CHB_OUT_F32) ? 4 : 2;
    const int64_t row0 = (int64_t)m_base + cr;
    char* cbase = reinterpret_cast<char*>(p.C) + (row0 * p.ldc + col) * ESZ;
    const int64_t cstep = p.ldc * ESZ;
    const char* rbase = (EPI == CHB_EPI_RESID) ? reinterpret_cast<const char*>(p.resid) + (row0 * p.ld_resid + col) * 4 : nullptr;
    const int64_t rstep = p.ld_resid * 4;
    char* abase = (EPI == CHB_EPI_GELU || EPI == CHB_EPI_DGELU) ? reinterpret_cast<char*>(p.aux) + (row0 * p.ld_aux + col) * 2 : nullptr;
    const int64_t astep = p.ld_aux * 2;
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    float* wr[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) wr[b] = stage + i * 64 + (((4 * b + g) ^ i) << 2);
    const float* rd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) rd[k] = stage + (4 * k + cr) * 64 + ((c4 ^ (4 * k + cr)) << 2);
    // The epilogue's own global loads (residual / saved gelu' / positional rows) run DEPTH tile rows ahead: the loads of row
    // a + DEPTH are issued before row a crosses LDS.  Measured (tools/ab_build.sh, depths 0-5 A/B in one session): no depth
    // changes the wall time of any shape by more than run-to-run noise (+-2 %) - the epilogue is not waiting on these loads.
    constexpr bool HAS_IN = (EPI == CHB_EPI_PATCH || EPI == CHB_EPI_RESID || EPI == CHB_EPI_DGELU);
    constexpr int DEPTH = (EPI == CHB_EPI_RESID) ? EPI_DEPTH_RESID : (EPI == CHB_EPI_DGELU) ? EPI_DEPTH_DGELU : 1;   // rows in flight ahead
    float4 r4buf[DEPTH + 1][4];
    uint2 a2buf[DEPTH + 1][4];
    auto load_inputs = [&](int a, float4 (&r4)[4], uint2 (&a2)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rr = a * 16 + 4 * k;
            const int row = m_base + rr + cr;
            const bool okk = colok && (!GUARD || row < p.M);
            if (EPI == CHB_EPI_PATCH) {
                const int bi = row / p.period, pp = row - bi * p.period;
                if (okk) r4[k] = *reinterpret_cast<const float4*>(p.resid + (int64_t)(p.n_special + pp) * p.ld_resid + col);
            }
            if (EPI == CHB_EPI_RESID && okk) r4[k] = *reinterpret_cast<const float4*>(rbase + rr * rstep);
            if (EPI == CHB_EPI_DGELU && okk) a2[k] = *reinterpret_cast<const uint2*>(abase + rr * astep);
        }
    };
    if (HAS_IN) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (A0 + d < A1) load_inputs(A0 + d, r4buf[d % (DEPTH + 1)], a2buf[d % (DEPTH + 1)]);
    }
#pragma unroll
    for (int a = A0; a < A1; ++a) {
        float4 (&r4)[4] = r4buf[(a - A0) % (DEPTH + 1)];
        uint2 (&a2)[4] = a2buf[(a - A0) % (DEPTH + 1)];
        if (HAS_IN && a + DEPTH < A1) load_inputs(a + DEPTH, r4buf[(a + DEPTH - A0) % (DEPTH + 1)], a2buf[(a + DEPTH - A0) % (DEPTH + 1)]);
        int64_t orow[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rr = a * 16 + 4 * k;                      // compile-time row offset inside the wave's 128 rows
            const int row = m_base + rr + cr;
            orow[k] = row;
            ok[k] = colok && (!GUARD || row < p.M);
            if (EPI == CHB_EPI_PATCH) {
                const int bi = row / p.period, pp = row - bi * p.period;
                orow[k] = (int64_t)bi * (p.period + p.n_special) + p.n_special + pp;
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) *reinterpret_cast<float4_t*>(wr[b]) = acc[a][b];
        float4_t t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = *reinterpret_cast<const float4_t*>(rd[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rr = a * 16 + 4 * k;
            float v[4] = {t[k][0] + bias.x, t[k][1] + bias.y, t[k][2] + bias.z, t[k][3] + bias.w};
            if (EPI == CHB_EPI_GELU) {
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) gelu_both(v[e], v[e], d[e]);
                uint2 der;
                der.x = pack_bf16x2(d[0], d[1]);
                der.y = pack_bf16x2(d[2], d[3]);
                if (ok[k]) *reinterpret_cast<uint2*>(abase + rr * astep) = der;
            } else if (EPI == CHB_EPI_DGELU) {
                v[0] *= bf16_to_f32((bf16_t)(a2[k].x & 0xffff));
                v[1] *= bf16_to_f32((bf16_t)(a2[k].x >> 16));
                v[2] *= bf16_to_f32((bf16_t)(a2[k].y & 0xffff));
                v[3] *= bf16_to_f32((bf16_t)(a2[k].y >> 16));
            } else if (EPI == CHB_EPI_RESID || EPI == CHB_EPI_PATCH) {
                if (EPI == CHB_EPI_PATCH) {
                    v[0] += r4[k].x; v[1] += r4[k].y; v[2] += r4[k].z; v[3] += r4[k].w;
                }
                if (p.drop_thr) {
                    const uint64_t e0 = (uint64_t)orow[k] * (uint64_t)p.N + (uint64_t)col;
                    bool k0, k1, k2, k3;
                    chb_keep2((uint32_t)(e0 >> 1), p.drop_key, p.drop_thr, k0, k1);
                    chb_keep2((uint32_t)(e0 >> 1) + 1u, p.drop_key, p.drop_thr, k2, k3);
                    v[0] = k0 ? v[0] * p.drop_scale : 0.0f;
                    v[1] = k1 ? v[1] * p.drop_scale : 0.0f;
                    v[2] = k2 ? v[2] * p.drop_scale : 0.0f;
                    v[3] = k3 ? v[3] * p.drop_scale : 0.0f;
                }
                if (EPI == CHB_EPI_RESID) {
                    v[0] += r4[k].x; v[1] += r4[k].y; v[2] += r4[k].z; v[3] += r4[k].w;
                }
            }
            char* dst = (EPI == CHB_EPI_PATCH) ? reinterpret_cast<char*>(p.C) + (orow[k] * p.ldc + col) * ESZ : cbase + rr * cstep;
            if (ok[k]) {
                if (OUT == CHB_OUT_F32) {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(dst) = o;
                }
                csum[0] += v[0]; csum[1] += v[1]; csum[2] += v[2]; csum[3] += v[3];
            }
            // keep the erf polynomial of one 4-element group from being interleaved with the next three: the wider schedule
            // needs more than the 256 registers a wave has here and spills (measured: 0.82 -> 0.73 ms on the fc1 GEMM)
            if (EPI == CHB_EPI_GELU) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (p.colsum) {   // this wave's 128 rows x 64 columns: fold the 4 row-groups (lanes cr) and add once per column
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            csum[e] += __shfl_xor(csum[e], 16, 64);
            csum[e] += __shfl_xor(csum[e], 32, 64);
        }
        if (cr == 0 && colok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(p.colsum + col + e, csum[e]);
        }
    }
}
// The same for full-tile builds (FAST): written for a generic number of columns per lane.
// WIDE (bf16 outputs whose rows are 16-byte aligned): a lane owns EIGHT consecutive columns (8 lanes per row, 8 rows per
// wave-instruction) so that bf16 rows and the saved gelu' rows also move as 16 bytes per lane - the store tail is bound by the
// number of store instructions, not by their bytes (cdna_hip_programming.md T21): 16 instead of 32 per wave and output tile.
template <int EPI, int OUT, bool GUARD, int A0, int A1, bool WIDE = true>
__device__ __forceinline__ void epilogue_staged_wide(const GemmParams& p, float* stage, int m_base, int n_base, float4_t (&acc)[8][4],
                                                int lane) {
    constexpr int CW = (WIDE && OUT == CHB_OUT_BF16) ? 8 : 4;   // columns per lane
    constexpr int NQ = CW / 4;                                   // 16-byte scratch chunks per lane and row
    constexpr int LPR = 64 / CW;                                 // lanes per row: 16 | 8
    constexpr int RPI = 64 / LPR;                                // rows per wave-instruction: 4 | 8
    constexpr int NK = 16 / RPI;                                 // instructions per 16-row MFMA tile row: 4 | 2
    const int g = lane >> 4, i = lane & 15;
    const int cr = lane / LPR, cx = lane % LPR;
    const int col = n_base + cx * CW;
    const bool colok = !GUARD || col < p.N;
    float bias[CW];
#pragma unroll
    for (int e = 0; e < CW; ++e) bias[e] = 0.f;
    if (p.bias && colok) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col + 4 * q);
            bias[4 * q] = b4.x; bias[4 * q + 1] = b4.y; bias[4 * q + 2] = b4.z; bias[4 * q + 3] = b4.w;
        }
    }
    // lane-constant pieces of every address (the per-(a,k) part is a compile-time row count times a uniform stride)
    constexpr int ESZ = (OUT == CHB_OUT_F32) ? 4 : 2;
    const int64_t row0 = (int64_t)m_base + cr;
    char* cbase = reinterpret_cast<char*>(p.C) + (row0 * p.ldc + col) * ESZ;
    const int64_t cstep = p.ldc * ESZ;
    const char* rbase = (EPI == CHB_EPI_RESID) ? reinterpret_cast<const char*>(p.resid) + (row0 * p.ld_resid + col) * 4 : nullptr;
    const int64_t rstep = p.ld_resid * 4;
    char* abase = (EPI == CHB_EPI_GELU || EPI == CHB_EPI_DGELU) ? reinterpret_cast<char*>(p.aux) + (row0 * p.ld_aux + col) * 2 : nullptr;
    const int64_t astep = p.ld_aux * 2;
    float csum[CW];
#pragma unroll
    for (int e = 0; e < CW; ++e) csum[e] = 0.f;
    float* wr[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) wr[b] = stage + i * 64 + (((4 * b + g) ^ i) << 2);
    // read-back: row RPI k + cr, scratch chunks NQ cx + q.  WIDE: a 16-lane group reads rows 2j (even chunks ^ row = one parity)
    // and 2j + 1 (the other parity) - 16 different chunks, conflict-free like the 4-column form.
    const float* rd[NK][NQ];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int q = 0; q < NQ; ++q) rd[k][q] = stage + (RPI * k + cr) * 64 + (((NQ * cx + q) ^ (RPI * k + cr)) << 2);
    // The epilogue's own global loads (residual / saved gelu' / positional rows) run DEPTH tile rows ahead: the loads of row
    // a + DEPTH are issued before row a crosses LDS.  Measured (tools/ab_build.sh, depths 0-5 A/B in one session): no depth
    // changes the wall time of any shape by more than run-to-run noise (+-2 %) - the epilogue is not waiting on these loads.
    constexpr bool HAS_IN = (EPI == CHB_EPI_PATCH || EPI == CHB_EPI_RESID || EPI == CHB_EPI_DGELU);
    constexpr int DEPTH = (EPI == CHB_EPI_RESID) ? EPI_DEPTH_RESID : (EPI == CHB_EPI_DGELU) ? EPI_DEPTH_DGELU_WIDE : 1;   // rows in flight ahead
    float4 r4buf[DEPTH + 1][NK][NQ];
    uint32_t a2buf[DEPTH + 1][NK][CW / 2];
    auto load_inputs = [&](int a, float4 (&r4)[NK][NQ], uint32_t (&a2)[NK][CW / 2]) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int rr = a * 16 + RPI * k;
            const int row = m_base + rr + cr;
            const bool okk = colok && (!GUARD || row < p.M);
            if (EPI == CHB_EPI_PATCH) {
                const int bi = row / p.period, pp = row - bi * p.period;
                if (okk) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) r4[k][q] = *reinterpret_cast<const float4*>(p.resid + (int64_t)(p.n_special + pp) * p.ld_resid + col + 4 * q);
                }
            }
            if (EPI == CHB_EPI_RESID && okk) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) r4[k][q] = *reinterpret_cast<const float4*>(rbase + rr * rstep + 16 * q);
            }
            if (EPI == CHB_EPI_DGELU && okk) {
                if (CW == 8) {
                    const uint4 t4 = *reinterpret_cast<const uint4*>(abase + rr * astep);
                    a2[k][0] = t4.x; a2[k][1] = t4.y; a2[k][CW / 2 - 2] = t4.z; a2[k][CW / 2 - 1] = t4.w;
                } else {
                    const uint2 t2 = *reinterpret_cast<const uint2*>(abase + rr * astep);
                    a2[k][0] = t2.x; a2[k][1] = t2.y;
                }
            }
        }
    };
    if (HAS_IN) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (A0 + d < A1) load_inputs(A0 + d, r4buf[d % (DEPTH + 1)], a2buf[d % (DEPTH + 1)]);
    }
#pragma unroll
    for (int a = A0; a < A1; ++a) {
        float4 (&r4)[NK][NQ] = r4buf[(a - A0) % (DEPTH + 1)];
        uint32_t (&a2)[NK][CW / 2] = a2buf[(a - A0) % (DEPTH + 1)];
        if (HAS_IN && a + DEPTH < A1) load_inputs(a + DEPTH, r4buf[(a + DEPTH - A0) % (DEPTH + 1)], a2buf[(a + DEPTH - A0) % (DEPTH + 1)]);
        int64_t orow[NK];
        bool ok[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int rr = a * 16 + RPI * k;                    // compile-time row offset inside the wave's 128 rows
            const int row = m_base + rr + cr;
            orow[k] = row;
            ok[k] = colok && (!GUARD || row < p.M);
            if (EPI == CHB_EPI_PATCH) {
                const int bi = row / p.period, pp = row - bi * p.period;
                orow[k] = (int64_t)bi * (p.period + p.n_special) + p.n_special + pp;
            }
        }
        float4_t t[NK][NQ];
#ifdef CHB_ABL_NOLDS      // ablation (wrong values): no trip through the LDS scratch
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int q = 0; q < NQ; ++q) t[k][q] = acc[a][(k * NQ + q) & 3];
#else
#pragma unroll
        for (int b = 0; b < 4; ++b) *reinterpret_cast<float4_t*>(wr[b]) = acc[a][b];
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int q = 0; q < NQ; ++q) t[k][q] = *reinterpret_cast<const float4_t*>(rd[k][q]);
#endif
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int rr = a * 16 + RPI * k;
            float v[CW];
#pragma unroll
            for (int e = 0; e < CW; ++e) v[e] = t[k][e >> 2][e & 3] + bias[e];
            if (EPI == CHB_EPI_GELU) {
                uint32_t der[CW / 2];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    float d[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) gelu_both(v[4 * q + e], v[4 * q + e], d[e]);
                    der[2 * q] = pack_bf16x2(d[0], d[1]);
                    der[2 * q + 1] = pack_bf16x2(d[2], d[3]);
                    // keep the erf polynomial of one 4-element group from being interleaved with the next: the wider schedule
                    // needs more than the 256 registers a wave has here and spills (measured: 0.82 -> 0.73 ms on the fc1 GEMM)
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ok[k]) {
                    if (CW == 8) *reinterpret_cast<uint4*>(abase + rr * astep) = make_uint4(der[0], der[1], der[CW / 2 - 2], der[CW / 2 - 1]);
                    else *reinterpret_cast<uint2*>(abase + rr * astep) = make_uint2(der[0], der[1]);
                }
            } else if (EPI == CHB_EPI_DGELU) {
#pragma unroll
                for (int e = 0; e < CW; e += 2) {
                    v[e] *= bf16_to_f32((bf16_t)(a2[k][e >> 1] & 0xffff));
                    v[e + 1] *= bf16_to_f32((bf16_t)(a2[k][e >> 1] >> 16));
                }
            } else if (EPI == CHB_EPI_RESID || EPI == CHB_EPI_PATCH) {
                if (EPI == CHB_EPI_PATCH) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) { v[4 * q] += r4[k][q].x; v[4 * q + 1] += r4[k][q].y; v[4 * q + 2] += r4[k][q].z; v[4 * q + 3] += r4[k][q].w; }
                }
                if (p.drop_thr) {
                    const uint64_t e0 = (uint64_t)orow[k] * (uint64_t)p.N + (uint64_t)col;
#pragma unroll
                    for (int e = 0; e < CW; e += 2) {
                        bool k0, k1;
                        chb_keep2((uint32_t)(e0 >> 1) + (uint32_t)(e >> 1), p.drop_key, p.drop_thr, k0, k1);
                        v[e] = k0 ? v[e] * p.drop_scale : 0.0f;
                        v[e + 1] = k1 ? v[e + 1] * p.drop_scale : 0.0f;
                    }
                }
                if (EPI == CHB_EPI_RESID) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) { v[4 * q] += r4[k][q].x; v[4 * q + 1] += r4[k][q].y; v[4 * q + 2] += r4[k][q].z; v[4 * q + 3] += r4[k][q].w; }
                }
            }
            char* dst = (EPI == CHB_EPI_PATCH) ? reinterpret_cast<char*>(p.C) + (orow[k] * p.ldc + col) * ESZ : cbase + rr * cstep;
#ifdef CHB_ABL_NOSTORE    // ablation: the output stores left out (the values still feed the column sums)
            if (ok[k] && p.colsum == reinterpret_cast<const float*>(1)) {
#else
            if (ok[k]) {
#endif
                if (OUT == CHB_OUT_F32) {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else if (CW == 8) {
                    *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[CW - 4], v[CW - 3]),
                                                               pack_bf16x2(v[CW - 2], v[CW - 1]));
                } else {
                    *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                }
#pragma unroll
                for (int e = 0; e < CW; ++e) csum[e] += v[e];
            }
        }
    }
    if (p.colsum) {   // this wave's 128 rows x 64 columns: fold the row groups (lanes cr) and add once per column
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            if (CW == 8) csum[e] += __shfl_xor(csum[e], 8, 64);
            csum[e] += __shfl_xor(csum[e], 16, 64);
            csum[e] += __shfl_xor(csum[e], 32, 64);
        }
        if (cr == 0 && colok) {
#pragma unroll
            for (int e = 0; e < CW; ++e) atomicAdd(p.colsum + col + e, csum[e]);
        }
    }
}

// Tile queue of the persistent NT kernel: per launch slot, one counter per XCD (the XCD's tile list is private to its workgroups).
// Zero at load time; the workgroup that draws an XCD's LAST ticket zeroes its counter again, so a slot is clean whenever its
// launch has ended.
constexpr int TILE_QUEUE_SLOTS = 64;
__device__ int g_tile_ctr[TILE_QUEUE_SLOTS][8];

struct TileWalk {
    int dyn;                        // tile queue on: tile j > 0 of this workgroup is li_tab[j & 3] (claimed), not slot + j*stride
    int li_tab[4];
    int start, cnt, slot, stride;   // this workgroup's tiles: start + slot + j*stride, j = 0.. while < cnt
    int tiles_n, ntk;
    // panel walk (bn > 0): the XCD owns m-tiles [mlo, mlo + mcnt) and sweeps them one bn-wide column of n-tiles at a time,
    // n fastest; the XCD's 32 concurrent tiles then form a (32/bn) x bn block, the smallest A + B footprint per round
    int mlo, mcnt, bn;
};

struct Cursor {   // (output tile, k-tile) position of a staging stream
    int j, kt, m0, n0;
    bool valid;
};

__device__ __forceinline__ void cursor_set(Cursor& c, const TileWalk& w, int j) {
    c.j = j;
    c.kt = 0;
    int li = w.slot + j * w.stride;
    if (w.dyn) {
        const int q = j & 3;
        li = q == 0 ? w.li_tab[0] : q == 1 ? w.li_tab[1] : q == 2 ? w.li_tab[2] : w.li_tab[3];
    }
    c.valid = li < w.cnt;
    if (w.bn > 0) {
        const int l = c.valid ? li : 0;
        const int colsz = w.mcnt * w.bn;
        const int ncols = (w.tiles_n + w.bn - 1) / w.bn;
        const int sc = min(l / colsz, ncols - 1);
        const int rem = l - sc * colsz;
        const int width = min(w.bn, w.tiles_n - sc * w.bn);
        const int tm = rem / width;
        c.m0 = (w.mlo + tm) * 256;
        c.n0 = (sc * w.bn + (rem - tm * width)) * 256;
        return;
    }
    const int v = w.start + (c.valid ? li : 0);
    const int tm = v / w.tiles_n;
    c.m0 = tm * 256;
    c.n0 = (v - tm * w.tiles_n) * 256;
}

__device__ __forceinline__ void cursor_next(Cursor& c, const TileWalk& w) {
    if (++c.kt == w.ntk) cursor_set(c, w, c.j + 1);
}

// stage one 128-row half-tile (16 glds instructions over 8 waves: 2 per wave)
__device__ __forceinline__ void stage_half(const bf16_t* __restrict__ g, int64_t ld, int row0, int max_row, int k0,
                                           bf16_t* lds_half, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int inst = j * 8 + wave;
        const int r = inst * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < max_row ? gr : max_row - 1;
        glds16(g + (int64_t)gr * ld + k0 + c * 8, lds_half + inst * 512);
    }
}

// one of the 16 instructions of a half-tile (8 rows), edge rows clamped
__device__ __forceinline__ void stage_piece_clamped(const bf16_t* __restrict__ g, int64_t ld, int row0, int max_row, int k0,
                                                    bf16_t* lds_half, int inst, int lane) {
    const int r = inst * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int gr = row0 + r;
    gr = gr < max_row ? gr : max_row - 1;
    glds16(g + (int64_t)gr * ld + k0 + c * 8, lds_half + inst * 512);
}

// The same with the address split into a wave-uniform base (SALU: tile row, k offset) and two lane-constant 32-bit byte offsets
// (row within the half-tile x leading dimension + swizzled chunk), for half-tiles that need no row clamp: one 64-bit add per
// LDS-DMA instruction instead of a clamp, two 32-bit multiplies and a 64-bit multiply-add per lane.
__device__ __forceinline__ void stage_half_fast(const char* __restrict__ base, const uint32_t (&off)[2], bf16_t* lds_half, int wave) {
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(reinterpret_cast<const bf16_t*>(base + off[j]), lds_half + (j * 8 + wave) * 512);
}

// one staging call of the persistent kernel.  FAST (M % 256 == 0 and N % 256 == 0: every tile is full) compiles the clamp
// path (and the guarded epilogue) out; otherwise the fast path is taken per half-tile when rows [row0, row0 + 128) all exist
template <bool FAST>
__device__ __forceinline__ void stage_half_any(const bf16_t* __restrict__ g, int64_t ld, int row0, int max_row, int k0,
                                               const uint32_t (&off)[2], bf16_t* lds_half, int wave, int lane) {
    if (FAST || row0 + 128 <= max_row) stage_half_fast(reinterpret_cast<const char*>(g + (int64_t)row0 * ld + k0), off, lds_half, wave);
    else stage_half(g, ld, row0, max_row, k0, lds_half, wave, lane);
}

__device__ __forceinline__ void tile_walk_init(TileWalk& w, const GemmParams& p) {
    const int nb = p.tiles_m * p.tiles_n;
    const int q = nb >> 3, r = nb & 7, x = blockIdx.x & 7;
    w.start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    w.cnt = q + (x < r ? 1 : 0);
    w.slot = blockIdx.x >> 3;
    w.stride = gridDim.x >> 3;
    w.tiles_n = p.tiles_n;
    w.ntk = p.K / BK;
    w.bn = 0; w.mlo = 0; w.mcnt = 0;
    w.dyn = 0;
    w.li_tab[0] = w.slot; w.li_tab[1] = w.li_tab[2] = w.li_tab[3] = 0x7fffffff;
    const int qm = p.tiles_m >> 3, rm = p.tiles_m & 7;
    // (32 / bn + bn) tiles of A + B per round is smallest near bn = sqrt(32); the panel walk is taken when the m-tiles split
    // over the XCDs with <= 1/16 imbalance and the footprint shrinks by a quarter or more (N = 3072 here: L2-side fetch
    // 1.7 GB -> 1.1 GB per launch, profiles/); narrower outputs keep the linear walk, which measured a few % faster there
    const int ncols = max(1, (int)((float)p.tiles_n / 5.66f + 0.5f));
    const int bn = (p.tiles_n + ncols - 1) / ncols;
    const bool pays = 4.0f * (32.0f / p.tiles_n + p.tiles_n) >= 5.0f * (32.0f / bn + bn);
    if (qm >= 16 && (p.walk_panel == 2 || (p.walk_panel == 1 && pays))) {
        w.mlo = x * qm + min(x, rm);
        w.mcnt = qm + (x < rm ? 1 : 0);
        w.cnt = w.mcnt * p.tiles_n;
        w.bn = bn;
    }
}

template <int EPI, int OUT, bool FAST = false>
__global__ void __launch_bounds__(512, 2) gemm_nt256_kernel(GemmParams p) {
    // 128 KiB operand ring [2][A0 A1 B0 B1][128 x 64] + 8 x 4 KiB wave-private epilogue scratch = the CU's whole 160 KiB
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 4 * 8192 + 8 * 2048];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wm = wave >> 2, wn = wave & 3;
    const int g = lane >> 4, i = lane & 15;

    TileWalk w;
    tile_walk_init(w, p);
    if (w.slot >= w.cnt) return;
    CLOCK_STAMP(0);
    // Tile queue (K >= 8 K-tiles): a workgroup's first tile is its static one, every later tile is the next ticket of its XCD's
    // counter.  A workgroup that starts late (its CU was held by another stream's kernel - a collective) then simply draws fewer
    // tickets: the launch loses that CU's share of the time it was away, not a whole static tile share at the end.
    // Tile j + 1 is claimed while tile j is young: wave 0 draws the ticket behind the barrier of K-step k0 = slot & 3; the vmcnt(0) of
    // K-step k0 + 1 covers its return, wave 0 then hands it over through LDS in front of that step's barrier and every wave picks it
    // up behind it - the staging streams need it from K-step ntk - 2 on.
    w.dyn = (p.queue_slot >= 0 && w.ntk >= 8) ? 1 : 0;
    const int q_k0 = w.slot & 3;      // the XCD's 32 workgroups run in step: spread their draws over four K-steps
    int* q_ctr = &g_tile_ctr[w.dyn ? p.queue_slot : 0][blockIdx.x & 7];
    int* q_lds = reinterpret_cast<int*>(smem + 2 * 4 * 8192);
    int q_ticket = 0;

    // lane-constant byte offsets of the two LDS-DMA instructions a wave issues per half-tile (row 64 j + 8 wave + lane / 8)
    uint32_t offa[2], offb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (j * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        offa[j] = (uint32_t)(((int64_t)r * p.lda + c * 8) * 2);
        offb[j] = (uint32_t)(((int64_t)r * p.ldb + c * 8) * 2);
    }
    // staging streams: A runs one step ahead of compute, B two steps ahead
    Cursor ca, cb, cc;
    cursor_set(cc, w, 0);
    cursor_set(ca, w, 0);
    cursor_set(cb, w, 0);
    // prologue: all of step 0, B halves of step 1
    stage_half_any<FAST>(p.A, p.lda, ca.m0, p.M, 0, offa, smem + 0 * 8192, wave, lane);
    stage_half_any<FAST>(p.A, p.lda, ca.m0 + 128, p.M, 0, offa, smem + 1 * 8192, wave, lane);
    stage_half_any<FAST>(p.B, p.ldb, cb.n0, p.N, 0, offb, smem + 2 * 8192, wave, lane);
    stage_half_any<FAST>(p.B, p.ldb, cb.n0 + 128, p.N, 0, offb, smem + 3 * 8192, wave, lane);
    cursor_next(ca, w);
    cursor_next(cb, w);
    if (cb.valid) {
        stage_half_any<FAST>(p.B, p.ldb, cb.n0, p.N, cb.kt * BK, offb, smem + (4 + 2) * 8192, wave, lane);
        stage_half_any<FAST>(p.B, p.ldb, cb.n0 + 128, p.N, cb.kt * BK, offb, smem + (4 + 3) * 8192, wave, lane);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    cursor_next(cb, w);
    __builtin_amdgcn_s_barrier();

    float4_t acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};

    // Lane-constant LDS element offsets.  A tile row r = base + 16*a + i has swizzle ((r >> 1) & 7) = (i >> 1) & 7 for
    // every a (16*a is a multiple of 16), so each fragment is one of four lane bases (A/B x k-step) plus a compile-time
    // row offset: all address arithmetic leaves the K-loop.
    const int sw = (i >> 1) & 7;
    const int a_off0 = i * 64 + (((0 + g) ^ sw) << 3), a_off1 = i * 64 + (((4 + g) ^ sw) << 3);
    const int b_off0 = ((wn & 1) * 64 + i) * 64 + (((0 + g) ^ sw) << 3), b_off1 = ((wn & 1) * 64 + i) * 64 + (((4 + g) ^ sw) << 3);

    PHASE_DECL
    for (int s = 0; cc.valid; ++s) {
        PHASE(0)       // cursor bookkeeping (and, once per tile, the epilogue: counted apart below)
        bf16_t* ring = smem + (s & 1) * 4 * 8192;
        bf16_t* nring = smem + ((s + 1) & 1) * 4 * 8192;
        const bf16_t* As = ring + wm * 8192;
        const bf16_t* Bs = ring + (2 + (wn >> 1)) * 8192;
        bf16x8_t af[4][2], bq[4][2];
        const bool last_k = cc.kt == w.ntk - 1;
        const bool interior = cc.m0 + 256 <= p.M && cc.n0 + 256 <= p.N;
        float* stage = reinterpret_cast<float*>(smem + 2 * 4 * 8192) + wave * 1024;

        // ---------------- phase 1: rows 0-63 of the wave x all 64 columns (32 MFMAs).
        // The other ring's A slots were last read in phase 2 of the previous step (before its barrier): restage them now.
        if (ca.valid) {
            stage_half_any<FAST>(p.A, p.lda, ca.m0, p.M, ca.kt * BK, offa, nring + 0 * 8192, wave, lane);
            stage_half_any<FAST>(p.A, p.lda, ca.m0 + 128, p.M, ca.kt * BK, offa, nring + 1 * 8192, wave, lane);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bq[b][ks] = *reinterpret_cast<const bf16x8_t*>(Bs + (ks ? b_off1 : b_off0) + b * 16 * 64);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[a][ks] = *reinterpret_cast<const bf16x8_t*>(As + (ks ? a_off1 : a_off0) + a * 16 * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        PHASE(1)       // A staging issue + 16 fragment reads + their wait
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[b][ks], af[a][ks], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        PHASE(2)       // 32 MFMAs issued

        // ---------------- phase 2: rows 64-127 (32 MFMAs).  No barrier separates the phases: the one below orders every read of
        // this ring (B and A rows 0-63 in phase 1, A rows 64-127 here) before the restaging that follows it.
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[a][ks] = *reinterpret_cast<const bf16x8_t*>(As + (ks ? a_off1 : a_off0) + (64 + a * 16) * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PHASE(3)       // 8 fragment reads + their wait
        // all of step s+1 (A issued in phase 1 of this step, B after the previous step's barrier) has landed when the barrier opens
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PHASE(4)       // vmcnt(0)
        if (w.dyn && cc.kt == q_k0 + 1 && wave == 0) {     // scalar branch
            // The ticket drawn behind the previous step's barrier has returned (that vmcnt(0) covers it).  The asm statement marks
            // the first point at which the register may be looked at: tools/check_ticket_isa.py verifies on the built code that no
            // instruction touches it between the atomic and a vmcnt wait.
            asm volatile("" : "+v"(q_ticket));
            if (lane == 0) {
                // tickets 0 .. cnt-1 are drawn per XCD and launch (one per tile started): ticket t is tile stride + t of the
                // XCD's list, and whoever holds the last one leaves the counter clean for the slot's next launch
                *q_lds = w.stride + q_ticket;
                if (q_ticket == w.cnt - 1) __hip_atomic_store(q_ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the word is in LDS before this wave reaches the barrier
        }
        __builtin_amdgcn_s_barrier();          // every wave has consumed this ring's A and B half-tiles; step s+1 has landed
        __builtin_amdgcn_sched_barrier(0);
        PHASE(5)       // barrier
        if (w.dyn) {
            // The next ticket is drawn right behind this step's vmcnt(0): its round trip has a whole K-step before the next vmcnt(0)
            // has to cover it.  (Inline asm: hipcc's atomicAdd aggregates over the wave and reads the result back at once - an
            // s_waitcnt vmcnt(0) right behind the atomic, which stalled wave 0 and with it the workgroup for the round trip; an
            // asm result is outside its bookkeeping.  Lane 0 only, in-place operand.)  The hand-off to the other waves goes
            // through a word of wave 0's epilogue scratch, idle during the K-loop: global stores / coherent loads would each
            // sit in a K-step's vmcnt(0) for longer than the LDS-DMA loads that wait is there for (measured: -2...-3 % per GEMM).
            if (cc.kt == q_k0 + 1) {               // wave 0's LDS write is one barrier old
                const int nxt = __builtin_amdgcn_readfirstlane(*q_lds);
                const int q = (cc.j + 1) & 3;
                if (q == 0) w.li_tab[0] = nxt; else if (q == 1) w.li_tab[1] = nxt; else if (q == 2) w.li_tab[2] = nxt; else w.li_tab[3] = nxt;
            }
            if (cc.kt == q_k0 && wave == 0) {      // scalar branch
                q_ticket = 1;
                uint64_t save;
                asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %[t], %[off], %[t], %[ptr] sc0\n\ts_mov_b64 exec, %[sv]"
                             : [t] "+v"(q_ticket), [sv] "=&s"(save)
                             : [off] "v"(0), [ptr] "s"(q_ctr)
                             : "memory");
            }
        }
        if (cb.valid) {                        // this ring's B slots are free now: step s+2 goes into them
            stage_half_any<FAST>(p.B, p.ldb, cb.n0, p.N, cb.kt * BK, offb, ring + 2 * 8192, wave, lane);
            stage_half_any<FAST>(p.B, p.ldb, cb.n0 + 128, p.N, cb.kt * BK, offb, ring + 3 * 8192, wave, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[4 + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[b][ks], af[a][ks], acc[4 + a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        PHASE(6)       // B staging issue + 32 MFMAs issued

        cursor_next(ca, w);
        cursor_next(cb, w);
        // ---------------- end of an output tile: epilogue (later steps' loads keep flying)
        if (last_k) {
            PHASE(7)
            if (FAST && EPI != CHB_EPI_DGELU) epilogue_staged_wide<EPI, OUT, false, 0, 8, true>(p, stage, cc.m0 + wm * 128, cc.n0 + wn * 64, acc, lane);   // (gelu'-multiply: the 4-column form with its inputs one row ahead is the faster one)
            else if (FAST || interior) epilogue_staged<EPI, OUT, false, 0, 8>(p, stage, cc.m0 + wm * 128, cc.n0 + wn * 64, acc, lane);
            else epilogue_staged<EPI, OUT, true, 0, 8>(p, stage, cc.m0 + wm * 128, cc.n0 + wn * 64, acc, lane);
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};
            PHASE(8)   // epilogue
#ifdef CHB_PHASE_STAMPS
            ph_sum[9] += 1;
#endif
        }
        cursor_next(cc, w);
    }
    PHASE_FLUSH
    CLOCK_STAMP(2);
}


// ---- NT, persistent 256x256 tiles, fragment reads software-pipelined under the MFMAs, staging spread over the K-step ----------
// Same geometry, ring and epilogue as gemm_nt256_kernel.  What tools/gemm_phases.py and the ablation builds (CHB_ABL_*) showed
// on the lockstep kernel: with the staging left out it runs 1.4x faster, with the fragment reads left out 1.07x - the cost is
// the LDS-DMA, and it is paid twice: (a) the 64 one-KiB instructions of a K-step all leave in one burst behind the barrier, the
// CU's address unit takes them one per ~16 cycles, and a wave queued there issues no MFMA; (b) the A operand (activations, new
// to the chip) has half a K-step to land.  This kernel therefore
//  * cuts the K-step into FOUR quarters of 16 MFMAs, ordered so that all of B and rows 0-63 of A are read out after the second:
//      Q0  read bK1 = B(k-half 1), aY = A(rows 0-63, k-half 1)        | acc[0..3] += bK0 x aX
//      Q1  read aX = A(rows 64-127, k-half 0) | lgkmcnt(4) | BARRIER 1 | acc[0..3] += bK1 x aY     + 3 pieces of step s+2
//      Q2  read aY = A(rows 64-127, k-half 1)                          | acc[4..7] += bK0 x aX     + 3 pieces
//          lgkmcnt(0) | counted vmcnt: step s+1 has landed | BARRIER 2
//      Q3  read bK0, aX of step s+1 (other ring)                       | acc[4..7] += bK1 x aY     + 2 pieces (A rows 64-127)
//    every quarter issues the NEXT quarter's fragment reads first (inline asm, hand-counted lgkmcnt - see below), so no ds_read
//    burst stands in front of an MFMA block; the same 64 fragment registers as the lockstep kernel, the same K order per
//    accumulator (bit-identical output);
//  * frees 48 of the ring's 64 KiB per step at barrier 1: six of a wave's eight pieces go out over Q1-Q2, one in front of a group
//    of four MFMAs, with more than a whole K-step to land; the step-end wait leaves exactly those six in flight (vmcnt(6));
//  * keeps wave-uniform byte pointers of the two staging streams, so a piece is an LDS-offset move, one 64-bit add and the
//    LDS-DMA instead of a 64-bit multiply-add chain.
// Measured steps on the way (tools/gemm_ab.py, bit-equal each): reads pipelined only -2..-4 % vs lockstep; + pointer streams and
// pieces between MFMAs +1..+3 %; pieces split Q3 / next Q0 (later, not earlier) -2..-5 %; an L2 touch of A two or four steps
// ahead -3..-5 %; staggering the two wave groups 0 %; barrier 1 with B early +3..+6 %.  On a tile's last K-step the Q3 read is
// left out and done after the epilogue instead, so the epilogue keeps its registers.
// Fragment reads of the pipelined kernel are inline asm with hand-counted waits.  hipcc cannot count them: an LDS-DMA
// (global_load_lds) in flight is a "flat access that may touch LDS" to its wait-count pass, and while one is pending it turns every
// lgkmcnt wait into lgkmcnt(0) - which here is always, by design.  The waits name the registers they make valid ("+v"), so no
// use of a fragment can be scheduled above its wait.
template <int OFF>
__device__ __forceinline__ void ds_read128(bf16x8_t& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
#define FRAG4(x) "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)LDS_PTR(p); }

template <int EPI, int OUT, bool FAST>
__global__ void __launch_bounds__(512, 2) gemm_nt256sp_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 4 * 8192 + 8 * 2048];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wm = wave >> 2, wn = wave & 3;
    const int g = lane >> 4, i = lane & 15;

    TileWalk w;
    tile_walk_init(w, p);
    if (w.slot >= w.cnt) return;
    // Tile queue as in gemm_nt256_kernel (K >= 8 K-tiles): the ticket for tile j+1 is drawn by wave 0 behind barrier 2 of K-step
    // k0 = slot & 3 of tile j; it is older than everything the step-end wait of K-step k0+1 leaves in flight, so it has returned
    // by then: wave 0 hands it over through word 0 of its idle epilogue scratch in front of that step's barrier 2, every wave picks
    // it up behind the barrier - the staging stream enters tile j+1 at K-step ntk-2.
    w.dyn = (p.queue_slot >= 0 && w.ntk >= 8) ? 1 : 0;
    const int q_k0 = w.slot & 3;
    int* q_ctr = &g_tile_ctr[w.dyn ? p.queue_slot : 0][blockIdx.x & 7];
    int* q_lds = reinterpret_cast<int*>(smem + 2 * 4 * 8192);
    int q_ticket = 0;

    uint32_t offa[2], offb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (j * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        offa[j] = (uint32_t)(((int64_t)r * p.lda + c * 8) * 2);
        offb[j] = (uint32_t)(((int64_t)r * p.ldb + c * 8) * 2);
    }
    Cursor cs, cc;                    // staging stream (two steps ahead of compute), compute
    cursor_set(cc, w, 0);
    cursor_set(cs, w, 0);
    // Staging stream: wave-uniform byte pointers to (tile row / column 0, current K-tile) of A and B, moved by one K-tile per step
    // and recomputed only when the stream enters a new output tile - a piece (one 1 KiB LDS-DMA instruction) then costs one
    // 64-bit add instead of a 64-bit multiply-add chain (tools/gemm_phases.py + ablations: the 64 pieces of a K-step cost the
    // lockstep kernel about as much wall time as a quarter of its MFMAs, half of it scalar address arithmetic in front of each).
    const char* pa = nullptr;
    const char* pb = nullptr;
    const int64_t a_half = (int64_t)128 * p.lda * 2, b_half = (int64_t)128 * p.ldb * 2;
    // (macros, not lambdas: a by-reference capture of the tile walk, which the tile queue writes to, sends it to scratch memory)
#define STREAM_SET() { pa = reinterpret_cast<const char*>(p.A + (int64_t)cs.m0 * p.lda + cs.kt * BK); \
                       pb = reinterpret_cast<const char*>(p.B + (int64_t)cs.n0 * p.ldb + cs.kt * BK); }
#define STREAM_NEXT() { cursor_next(cs, w); if (cs.kt == 0) STREAM_SET() else { pa += BK * 2; pb += BK * 2; } }
    // piece e of a step: e = 0..3 -> B (half e >> 1, instruction e & 1), e = 4..7 -> A; instruction 0 of a half = its rows 0-63
    auto piece = [&](int e, bf16_t* ring_) {
        const int h = (e >> 1) & 1, j = e & 1;
        if (e < 4) {
            if (FAST || cs.n0 + h * 128 + 128 <= p.N) glds16(reinterpret_cast<const bf16_t*>(pb + h * b_half + offb[j]), ring_ + (2 + h) * 8192 + (j * 8 + wave) * 512);
            else stage_piece_clamped(p.B, p.ldb, cs.n0 + h * 128, p.N, cs.kt * BK, ring_ + (2 + h) * 8192, j * 8 + wave, lane);
        } else {
#ifndef CHB_ABL_HALF
            if (FAST || cs.m0 + h * 128 + 128 <= p.M) glds16(reinterpret_cast<const bf16_t*>(pa + h * a_half + offa[j]), ring_ + h * 8192 + (j * 8 + wave) * 512);
            else stage_piece_clamped(p.A, p.lda, cs.m0 + h * 128, p.M, cs.kt * BK, ring_ + h * 8192, j * 8 + wave, lane);
#endif
        }
    };
    auto stage_step = [&](bf16_t* ring_) {
#pragma unroll
        for (int e = 0; e < 8; ++e) piece(e, ring_);
    };
    STREAM_SET()
    stage_step(smem);
    STREAM_NEXT()
    if (cs.valid) stage_step(smem + 4 * 8192);
    if (cs.valid) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // a piece is ONE instruction in every build (ragged ones clamp rows)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STREAM_NEXT()
    __builtin_amdgcn_s_barrier();

    float4_t acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int sw = (i >> 1) & 7;
    const int a_off0 = i * 64 + (((0 + g) ^ sw) << 3), a_off1 = i * 64 + (((4 + g) ^ sw) << 3);
    const int b_off0 = ((wn & 1) * 64 + i) * 64 + (((0 + g) ^ sw) << 3), b_off1 = ((wn & 1) * 64 + i) * 64 + (((4 + g) ^ sw) << 3);
    bf16x8_t aX[4], aY[4], bK0[4], bK1[4];
    // byte addresses of the four lane bases in ring 0 (A / B x k-half); fragment t of a set is +2 KiB t, rows 64-127 +8 KiB
    const uint32_t a_adr0 = lds_addr(smem + wm * 8192 + a_off0), a_adr1 = lds_addr(smem + wm * 8192 + a_off1);
    const uint32_t b_adr0 = lds_addr(smem + (2 + (wn >> 1)) * 8192 + b_off0), b_adr1 = lds_addr(smem + (2 + (wn >> 1)) * 8192 + b_off1);
#ifdef CHB_ABL_NOREAD
#define READ4(x, adr, base) { asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(adr)); }
#else
#define READ4(x, adr, base) { ds_read128<(base)>(x[0], adr); ds_read128<(base) + 2048>(x[1], adr); ds_read128<(base) + 4096>(x[2], adr); ds_read128<(base) + 6144>(x[3], adr); }
#endif
    READ4(bK0, b_adr0, 0)
    READ4(aX, a_adr0, 0)
    constexpr int EPI_OPS = !FAST ? 0 : (EPI == CHB_EPI_NONE) ? 16 : (EPI == CHB_EPI_GELU || EPI == CHB_EPI_DGELU) ? 32 : 63;

    PHASE_DECL
    for (int s = 0; cc.valid; ++s) {
        bf16_t* ring = smem + (s & 1) * 4 * 8192;
        const uint32_t ro = (s & 1) * 65536, nro = 65536 - ro;          // ring byte offsets: this step's, the next step's
        const uint32_t aA0 = a_adr0 + ro, aA1 = a_adr1 + ro, bA1 = b_adr1 + ro;
        const uint32_t naA0 = a_adr0 + nro, nbA0 = b_adr0 + nro;
        const bool last_k = cc.kt == w.ntk - 1;
        const bool first_k = cc.kt == 0 && s > 0;
        const bool interior = cc.m0 + 256 <= p.M && cc.n0 + 256 <= p.N;
        float* stage = reinterpret_cast<float*>(smem + 2 * 4 * 8192) + wave * 1024;

        // ---- Q0   (in flight on entry: bK0, aX = A rows 0-63 k-half 0: 8 reads)
        READ4(bK1, bA1, 0)
        READ4(aY, aA1, 0)
        asm volatile("s_waitcnt lgkmcnt(8)" : FRAG4(bK0), FRAG4(aX));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bK0[b], aX[a], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- Q1
        READ4(aX, aA0, 8192)
        asm volatile("s_waitcnt lgkmcnt(4)" : FRAG4(bK1), FRAG4(aY) : : "memory");
        // barrier 1: every wave has read ALL of B and rows 0-63 of both A halves of this ring - those 48 KiB take step s+2 now,
        // one piece in front of a group of four MFMAs, over Q1 and Q2
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool staging = cs.valid;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#ifndef CHB_ABL_NOSTAGE
            if (staging && a < 3) piece(a == 2 ? 4 : a, ring);
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bK1[b], aY[a], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- Q2
        READ4(aY, aA1, 8192)
        asm volatile("s_waitcnt lgkmcnt(4)" : FRAG4(aX));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#ifndef CHB_ABL_NOSTAGE
            if (staging && a < 3) piece(a == 2 ? 6 : 2 + a, ring);
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[4 + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bK0[b], aX[a], acc[4 + a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- barrier 2: this ring is read out, step s+1 has landed.  The counter retires in issue order: what may stay in flight
        // is what was issued AFTER the last LDS-DMA of step s+1 - this step's six early pieces and, behind a tile's epilogue, its
        // loads and stores.
        asm volatile("s_waitcnt lgkmcnt(0)" : FRAG4(aY) : : "memory");
        PHASE(1)       // Q3 of the previous step, bookkeeping, Q0, Q1, Q2 and the last reads' wait
        {
            const int young = (staging ? 6 : 0) + (first_k ? EPI_OPS : 0);     // ragged builds: EPI_OPS = 0 (a guarded epilogue may skip stores)
            if (young >= 63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
            else if (young == 38) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
            else if (young == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if (young == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
            else if (young == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (young == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#ifdef CHB_PHASE_STAMPS   // the step-end wait by position in the tile: K-steps 0, 1, 2 apart from the rest
        if (cc.kt == 0) { PHASE(3) } else if (cc.kt == 1) { PHASE(2) } else if (cc.kt == 2) { PHASE(6) } else { PHASE(4) }
#endif
        if (w.dyn && cc.kt == q_k0 + 1 && wave == 0) {     // scalar branch
            asm volatile("" : "+v"(q_ticket));             // first look at the ticket register: behind the counted wait above (tools/check_ticket_isa.py)
            if (lane == 0) {
                // tickets 0 .. cnt-1 are drawn per XCD and launch (one per tile started): ticket t is tile stride + t of the XCD's
                // list, and whoever holds the last one leaves the counter clean for the slot's next launch
                *q_lds = w.stride + q_ticket;
                if (q_ticket == w.cnt - 1) __hip_atomic_store(q_ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the word is in LDS before this wave reaches the barrier
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PHASE(5)       // barrier
        if (w.dyn) {
            if (cc.kt == q_k0 && wave == 0) {      // scalar branch; lane 0 only, in-place operand (see gemm_nt256_kernel)
                q_ticket = 1;
                uint64_t save;
                asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %[t], %[off], %[t], %[ptr] sc0\n\ts_mov_b64 exec, %[sv]"
                             : [t] "+v"(q_ticket), [sv] "=&s"(save)
                             : [off] "v"(0), [ptr] "s"(q_ctr)
                             : "memory");
            }
            if (cc.kt == q_k0 + 1) {
                const int nxt = __builtin_amdgcn_readfirstlane(*q_lds);
                const int q = (cc.j + 1) & 3;
                if (q == 0) w.li_tab[0] = nxt; else if (q == 1) w.li_tab[1] = nxt; else if (q == 2) w.li_tab[2] = nxt; else w.li_tab[3] = nxt;
            }
        }
        // ---- Q3
        if (!last_k) {
            READ4(bK0, nbA0, 0)
            READ4(aX, naA0, 0)
        }
        // rows 64-127 of both A halves of step s+2
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#ifndef CHB_ABL_NOSTAGE
            if (staging && a < 2) piece(5 + 2 * a, ring);
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[4 + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bK1[b], aY[a], acc[4 + a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (staging) STREAM_NEXT()

        if (last_k) {
            PHASE(7)
            if (FAST && EPI != CHB_EPI_DGELU) epilogue_staged_wide<EPI, OUT, false, 0, 8, true>(p, stage, cc.m0 + wm * 128, cc.n0 + wn * 64, acc, lane);   // (gelu'-multiply: the 4-column form with its inputs one row ahead is the faster one)
            else if (FAST || interior) epilogue_staged<EPI, OUT, false, 0, 8>(p, stage, cc.m0 + wm * 128, cc.n0 + wn * 64, acc, lane);
            else epilogue_staged<EPI, OUT, true, 0, 8>(p, stage, cc.m0 + wm * 128, cc.n0 + wn * 64, acc, lane);
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};
            // the next tile's first fragments (its step 0 landed before this step's barrier)
            READ4(bK0, nbA0, 0)
            READ4(aX, naA0, 0)
            PHASE(8)   // epilogue
#ifdef CHB_PHASE_STAMPS
            ph_sum[9] += 1;
#endif
        }
        cursor_next(cc, w);
    }
    PHASE_FLUSH
}
#undef READ4
#undef STREAM_SET
#undef STREAM_NEXT


// ---- NT, persistent 256x256 tiles, two wave groups in anti-phase ("ping-pong") ----------------------------------------------
// Same geometry as gemm_nt256_kernel (8 waves = 2(M) x 4(N), 128x64 outputs per wave, 64-deep K-tiles, 128 KiB ring + 32 KiB
// epilogue scratch), different clock-work: a K-tile is FOUR phases of 16 MFMAs (one quadrant of the wave's outputs each), every
// phase is  { ds_read this quadrant's fragments | issue one 16 KiB staging element | counted vmcnt | barrier | lgkmcnt(0) |
// 16 MFMAs | barrier },  and the waves of group 1 (rows 128-255) run ONE BARRIER behind those of group 0.  A SIMD hosts one wave
// of each group, so while one reads LDS and issues DMA the other owns the MFMA pipe, and they swap every barrier - the pipe no
// longer waits for a CU-wide fragment-read burst (gemm_nt256_kernel: both waves of a SIMD leave its barrier together and want
// the same unit).
//
// Staging elements are cut so that each is READ IN EXACTLY ONE PHASE (16 KiB = 128 rows x 64 k, two LDS-DMA instructions per wave):
//   e0 = A rows {128 grp + r, r < 64}        (both groups' first row halves)    read in phase 0
//   e1 = B rows {64 wn + r, r < 32}          (every wave's first 32 columns)    read in phase 0, fragments kept for phase 3
//   e2 = B rows {64 wn + 32 + r}                                               read in phase 1, kept for phase 2
//   e3 = A rows {128 grp + 64 + r}                                             read in phase 2
// Stream element h = 4 T + e (T = K-tile counter across output tiles) lives in ring slot h & 7.  In phase c = 4 T + p a wave
// issues element c + 6 and waits vmcnt(8): all but the four youngest elements have landed, i.e. everything up to c + 2, which is
// what phase c + 1 reads (RAW: a read comes one phase after the wait that retires its element, with a barrier between; WAR:
// element h overwrites h - 8, last read in phase h - 8 or h - 9, at least two phases - four barriers - earlier, so the lagging
// group's reads have retired too).  Epilogue stores share the vmcnt counter: the epilogue ends with a wait that leaves only ITS
// operations outstanding (so every element issued before it has landed) and the first K-tile of the next output tile, which
// reads only those elements, waits for nothing.
// Full tiles only (M % 256 == 0, N % 256 == 0) and K >= 4 K-tiles; launch_nt falls back to gemm_nt256_kernel otherwise.
template <int EPI, int OUT>
__global__ void __launch_bounds__(512, 2) gemm_nt256pp_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[8 * 8192 + 8 * 2048];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int grp = wave >> 2, wn = wave & 3;
    const int g = lane >> 4, i = lane & 15;

    TileWalk w;
    tile_walk_init(w, p);
    if (w.slot >= w.cnt) return;
    const int nmy = (w.cnt - w.slot + w.stride - 1) / w.stride;
    const int total = nmy * w.ntk;

    // lane-constant byte offsets of the two LDS-DMA instructions of an element: element row r = 8 (8 j + wave) + lane / 8
    uint32_t offa[2], offb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (j * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        offa[j] = (uint32_t)(((int64_t)((r >> 6) * 128 + (r & 63)) * p.lda + c * 8) * 2);
        offb[j] = (uint32_t)(((int64_t)((r >> 5) * 64 + (r & 31)) * p.ldb + c * 8) * 2);
    }
    // The staging stream runs six elements ahead of the compute phase: phase p of K-tile T issues element p + 2 of the stream's
    // K-tile (T + 1 for p = 0, 1; T + 2 for p = 2, 3), so which element a phase stages is a compile-time fact and the stream's
    // two row pointers move once per K-tile (scalar work per phase: one add and the LDS slot).
    Cursor cs, cc;
    cursor_set(cs, w, 0);
    cursor_set(cc, w, 0);
    const char* sa = nullptr;     // stream K-tile: A rows m0.., B rows n0.. at its k offset
    const char* sb = nullptr;
    auto stream_ptrs = [&]() {
        sa = reinterpret_cast<const char*>(p.A + (int64_t)cs.m0 * p.lda + cs.kt * BK);
        sb = reinterpret_cast<const char*>(p.B + (int64_t)cs.n0 * p.ldb + cs.kt * BK);
    };
    const int64_t a_half = (int64_t)64 * p.lda * 2, b_half = (int64_t)32 * p.ldb * 2;      // bytes to e3 / e2 from e0 / e1
    // element E (compile time) of the stream's K-tile into ring slot `slot`
#define CHB_PP_STAGE(E, slot)                                                                                     \
    {                                                                                                             \
        bf16_t* dst_ = smem + (slot) * 8192;                                                                       \
        if ((E) == 0) stage_half_fast(sa, offa, dst_, wave);                                                      \
        else if ((E) == 3) stage_half_fast(sa + a_half, offa, dst_, wave);                                        \
        else if ((E) == 1) stage_half_fast(sb, offb, dst_, wave);                                                 \
        else stage_half_fast(sb + b_half, offb, dst_, wave);                                                      \
    }
    stream_ptrs();
    CHB_PP_STAGE(0, 0) CHB_PP_STAGE(1, 1) CHB_PP_STAGE(2, 2) CHB_PP_STAGE(3, 3)
    cursor_next(cs, w);          // K >= 4 K-tiles: the stream's second K-tile exists
    stream_ptrs();
    CHB_PP_STAGE(0, 4) CHB_PP_STAGE(1, 5)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind from here on

    float4_t acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int sw = (i >> 1) & 7;
    const int a_off0 = (grp * 64 + i) * 64 + (((0 + g) ^ sw) << 3), a_off1 = (grp * 64 + i) * 64 + (((4 + g) ^ sw) << 3);
    const int b_off0 = (wn * 32 + i) * 64 + (((0 + g) ^ sw) << 3), b_off1 = (wn * 32 + i) * 64 + (((4 + g) ^ sw) << 3);
    float* stage = reinterpret_cast<float*>(smem + 8 * 8192) + wave * 1024;
    bf16x8_t af[4][2], bq[4][2];

#define CHB_PP_SYNC_MFMA(A0, B0, E, SLOT)                                                                                            \
    {                                                                                                                             \
        const bool staged = cs.valid;                                                                                             \
        if (staged) CHB_PP_STAGE(E, SLOT)                                                                                         \
        if (!first_kt) {                                                                                                          \
            if (staged) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                          \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                 \
        }                                                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                          \
            _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                                         \
                _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                     \
                    acc[A0 + a][B0 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[B0 + b][ks], af[a][ks], acc[A0 + a][B0 + b], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                                             \
    }

    for (int T = 0; T < total; ++T) {
        const bf16_t* ring = smem + (T & 1) * 4 * 8192;
        const bool first_kt = cc.kt == 0;
        const bool last_k = cc.kt == w.ntk - 1;
        // ---- phase 0: rows 0-63 x columns 0-31
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bq[b][ks] = *reinterpret_cast<const bf16x8_t*>(ring + 1 * 8192 + (ks ? b_off1 : b_off0) + b * 16 * 64);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[a][ks] = *reinterpret_cast<const bf16x8_t*>(ring + 0 * 8192 + (ks ? a_off1 : a_off0) + a * 16 * 64);
        CHB_PP_SYNC_MFMA(0, 0, 2, ((T + 1) & 1) * 4 + 2)
        // ---- phase 1: rows 0-63 x columns 32-63
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bq[2 + b][ks] = *reinterpret_cast<const bf16x8_t*>(ring + 2 * 8192 + (ks ? b_off1 : b_off0) + b * 16 * 64);
        CHB_PP_SYNC_MFMA(0, 2, 3, ((T + 1) & 1) * 4 + 3)
        // the stream moves on to its next K-tile (elements 0 and 1 of it go out in phases 2 and 3)
        if (cs.valid) {
            cursor_next(cs, w);
            if (cs.valid) stream_ptrs();
        }
        // ---- phase 2: rows 64-127 x columns 32-63
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[a][ks] = *reinterpret_cast<const bf16x8_t*>(ring + 3 * 8192 + (ks ? a_off1 : a_off0) + a * 16 * 64);
        CHB_PP_SYNC_MFMA(4, 2, 0, (T & 1) * 4 + 0)
        // ---- phase 3: rows 64-127 x columns 0-31 (fragments already in registers)
        CHB_PP_SYNC_MFMA(4, 0, 1, (T & 1) * 4 + 1)

        if (last_k) {
            epilogue_staged_wide<EPI, OUT, false, 0, 8, true>(p, stage, cc.m0 + grp * 128, cc.n0 + wn * 64, acc, lane);
            // leave only (at most) the epilogue's own memory operations outstanding: every staging element issued before it has landed
            constexpr int EPI_OPS = (EPI == CHB_EPI_NONE) ? 32 : 63;
            if (EPI_OPS == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};
        }
        cursor_next(cc, w);
    }
#undef CHB_PP_SYNC_MFMA
#undef CHB_PP_STAGE
    if (grp == 0) __builtin_amdgcn_s_barrier();      // pairs with group 1's extra barrier
}

// ---- TN (wgrad): operand tiles are [64 m rows][128 cols] (256-byte rows, 16 chunks) ----------
// Fragments are transposed reads (ds_read_b64_tr_b16): within one instruction a 32-lane half
// touches rows {r0..r0+3, r0+8..r0+11}; XOR the chunk index with f(r) << 1,
// f(r) = (r & 3) | (((r >> 3) & 1) << 2), so those 8 rows cover all 64 banks once.
__device__ __forceinline__ int swz_tn(int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; }

__device__ __forceinline__ void stage_tn(const bf16_t* __restrict__ g, int64_t ld, int m0, int col0, int ncols,
                                         bf16_t* lds_tile, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int inst = j * 4 + wave;           // 16 instructions of 4 rows each
        const int r = inst * 4 + (lane >> 4);
        const int c = (lane & 15) ^ swz_tn(r);
        int gc = col0 + c * 8;
        gc = gc < ncols ? gc : ncols - 8;        // column edge: re-read a valid chunk, masked at store
        glds16(g + (int64_t)(m0 + r) * ld + gc, lds_tile + inst * 512);
    }
}

// 8 reduction-consecutive bf16 for MFMA lane (g, i): rows mb+8g..mb+8g+7 of column c0+i
__device__ __forceinline__ bf16x8_t frag_tn(const bf16_t* lds_tile, int mb, int c0, int g, int i) {
    const int q = i >> 2, pp = i & 3;
    const int chunk = (c0 >> 3) + (pp >> 1);
    const int r0 = mb + 8 * g + q, r1 = r0 + 4;
    const bf16_t* a0 = lds_tile + r0 * 128 + ((chunk ^ swz_tn(r0)) << 3) + 4 * (pp & 1);
    const bf16_t* a1 = lds_tile + r1 * 128 + ((chunk ^ swz_tn(r1)) << 3) + 4 * (pp & 1);
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)a0);
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)a1);
    short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

// the same fragment from a precomputed lane address: rows +0..3 at p, rows +4..7 at p + 4 rows
// the same pair of transposed reads as inline asm (see gemm_nt256sp_kernel: reads the compiler cannot count are counted by hand)
template <int OFF>
__device__ __forceinline__ void tr_read128(bf16x8_t& d, uint32_t addr) {
    short4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 1024));
    const short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    d = __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t frag_tn_at(const bf16_t* p) {
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)p);
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(p + 4 * 128));
    short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

struct TnParams {
    const bf16_t* X; int64_t ldx;
    const bf16_t* Y; int64_t ldy;
    float* W; int64_t ldw;
    int M, Kd, Nd;
    int tiles_k, tiles_n, splits, steps_per_split;
};

__global__ void __launch_bounds__(256) gemm_tn_kernel(TnParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * TILE_ELEMS];  // X0 Y0 X1 Y1
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int tiles = p.tiles_k * p.tiles_n;
    const int split = blockIdx.x / tiles;
    const int t = blockIdx.x - split * tiles;
    const int tk = t / p.tiles_n, tn = t - tk * p.tiles_n;
    const int k0 = tk * 128, n0 = tn * 128;
    const int wk = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, i = lane & 15;

    const int total_steps = p.M / 64;
    const int s_begin = split * p.steps_per_split;
    int s_end = s_begin + p.steps_per_split;
    s_end = s_end < total_steps ? s_end : total_steps;
    if (s_begin >= s_end) return;  // uniform per workgroup

    float4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};

    stage_tn(p.X, p.ldx, s_begin * 64, k0, p.Kd, smem, wave, lane);
    stage_tn(p.Y, p.ldy, s_begin * 64, n0, p.Nd, smem + TILE_ELEMS, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int s = s_begin; s < s_end; ++s) {
        const int cur = (s - s_begin) & 1;
        bf16_t* Xs = smem + cur * 2 * TILE_ELEMS;
        bf16_t* Ys = Xs + TILE_ELEMS;
        if (s + 1 < s_end) {
            bf16_t* Xn = smem + (cur ^ 1) * 2 * TILE_ELEMS;
            stage_tn(p.X, p.ldx, (s + 1) * 64, k0, p.Kd, Xn, wave, lane);
            stage_tn(p.Y, p.ldy, (s + 1) * 64, n0, p.Nd, Xn + TILE_ELEMS, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t xf[4], yf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) xf[a] = frag_tn(Xs, ks * 32, wk * 64 + a * 16, g, i);
#pragma unroll
            for (int b = 0; b < 4; ++b) yf[b] = frag_tn(Ys, ks * 32, wn * 64 + b * 16, g, i);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    // D[row = nd][col = kd]: lane holds 4 consecutive nd for one kd
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[b], xf[a], acc[a][b], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int kd = k0 + wk * 64 + a * 16 + i;
        if (kd < p.Kd) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int nd = n0 + wn * 64 + b * 16 + g * 4;
                float* dst = p.W + (int64_t)kd * p.ldw + nd;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nd + r < p.Nd) atomicAdd(dst + r, acc[a][b][r]);
            }
        }
    }
}

// ---- NT, persistent 128x256 tiles, TWO independent 256-thread workgroups per CU ------------------
// Each workgroup (4 waves = 1(M) x 4(N), 128x64 outputs per wave) keeps its own 3-deep ring of 32-deep
// K-tiles (A 128x32 + B 256x32 = 24 KiB per stage, 72 KiB) plus a 2 KiB/wave epilogue scratch: 80 KiB, so
// two workgroups share a CU (2 waves per SIMD from DIFFERENT workgroups).  One barrier per K-step, DMA two
// K-steps ahead (counted vmcnt), and - the point of this variant - while one workgroup is in its epilogue
// (VALU + stores) the other keeps the MFMA pipe busy.
// LDS rows are 64 bytes (4 chunks of 16 B); bank swizzle chunk ^ f(row), f = {0,2,3,1}[(row >> 2) & 3].
__device__ __forceinline__ int swz32(int r) { return (0x78 >> (((r >> 2) & 3) * 2)) & 3; }   // {0,2,3,1}

__device__ __forceinline__ void stage_k32(const bf16_t* __restrict__ g, int64_t ld, int row0, int max_row, int k0, int ninst,
                                          bf16_t* lds_tile, int wave, int lane) {
    // ninst/4 instructions per wave, each 16 rows x 64 B
    for (int j = 0; j < ninst / 4; ++j) {
        const int inst = j * 4 + wave;
        const int r = inst * 16 + (lane >> 2);
        const int c = (lane & 3) ^ swz32(r);
        int gr = row0 + r;
        gr = gr < max_row ? gr : max_row - 1;
        glds16(g + (int64_t)gr * ld + k0 + c * 8, lds_tile + inst * 512);
    }
}

__device__ __forceinline__ bf16x8_t frag_k32(const bf16_t* lds_tile, int r, int g) {
    return *reinterpret_cast<const bf16x8_t*>(lds_tile + r * 32 + ((g ^ swz32(r)) << 3));
}

template <int EPI, int OUT, bool GUARD>
__device__ __forceinline__ void epilogue_staged32(const GemmParams& p, float* stage, int m_base, int n_base, float4_t (&acc)[8][4],
                                                  int lane) {
    // scratch: 16 rows x 32 fp32 columns (one half of a 16x64 MFMA tile row); lane = 8 rows x 8 chunks of 4 columns
    const int g = lane >> 4, i = lane & 15;
    const int cr = lane >> 3, c4 = lane & 7;
    constexpr int ESZ = (OUT == CHB_OUT_F32) ? 4 : 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = n_base + 32 * h + c4 * 4;
        const bool colok = !GUARD || col < p.N;
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && colok) bias = *reinterpret_cast<const float4*>(p.bias + col);
        const int64_t row0 = (int64_t)m_base + cr;
        char* cbase = reinterpret_cast<char*>(p.C) + (row0 * p.ldc + col) * ESZ;
        const int64_t cstep = p.ldc * ESZ;
        const char* rbase = (EPI == CHB_EPI_RESID) ? reinterpret_cast<const char*>(p.resid) + (row0 * p.ld_resid + col) * 4 : nullptr;
        const int64_t rstep = p.ld_resid * 4;
        char* abase = (EPI == CHB_EPI_GELU || EPI == CHB_EPI_DGELU) ? reinterpret_cast<char*>(p.aux) + (row0 * p.ld_aux + col) * 2 : nullptr;
        const int64_t astep = p.ld_aux * 2;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            float4 r4[2];
            uint2 a2[2];
            int64_t orow[2];
            bool ok[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int rr = a * 16 + 8 * k;
                const int row = m_base + rr + cr;
                orow[k] = row;
                ok[k] = colok && (!GUARD || row < p.M);
                if (EPI == CHB_EPI_PATCH) {
                    const int bi = row / p.period, pp = row - bi * p.period;
                    orow[k] = (int64_t)bi * (p.period + p.n_special) + p.n_special + pp;
                    if (ok[k]) r4[k] = *reinterpret_cast<const float4*>(p.resid + (int64_t)(p.n_special + pp) * p.ld_resid + col);
                }
                if (EPI == CHB_EPI_RESID && ok[k]) r4[k] = *reinterpret_cast<const float4*>(rbase + rr * rstep);
                if (EPI == CHB_EPI_DGELU && ok[k]) a2[k] = *reinterpret_cast<const uint2*>(abase + rr * astep);
            }
            // tile row a, column half h: this lane holds acc[a][2h + b'] = row i, columns 16 b' + 4g .. +3 of the half
#pragma unroll
            for (int b = 0; b < 2; ++b) *reinterpret_cast<float4_t*>(stage + i * 32 + ((((4 * b + g) ^ i) & 7) << 2)) = acc[a][2 * h + b];
            float4_t t[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) t[k] = *reinterpret_cast<const float4_t*>(stage + (8 * k + cr) * 32 + (((c4 ^ (8 * k + cr)) & 7) << 2));
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int rr = a * 16 + 8 * k;
                float v[4] = {t[k][0] + bias.x, t[k][1] + bias.y, t[k][2] + bias.z, t[k][3] + bias.w};
                if (EPI == CHB_EPI_GELU) {
                    float d[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) gelu_both(v[e], v[e], d[e]);
                    uint2 der;
                    der.x = pack_bf16x2(d[0], d[1]);
                    der.y = pack_bf16x2(d[2], d[3]);
                    if (ok[k]) *reinterpret_cast<uint2*>(abase + rr * astep) = der;
                } else if (EPI == CHB_EPI_DGELU) {
                    v[0] *= bf16_to_f32((bf16_t)(a2[k].x & 0xffff));
                    v[1] *= bf16_to_f32((bf16_t)(a2[k].x >> 16));
                    v[2] *= bf16_to_f32((bf16_t)(a2[k].y & 0xffff));
                    v[3] *= bf16_to_f32((bf16_t)(a2[k].y >> 16));
                } else if (EPI == CHB_EPI_RESID || EPI == CHB_EPI_PATCH) {
                    if (EPI == CHB_EPI_PATCH) {
                        v[0] += r4[k].x; v[1] += r4[k].y; v[2] += r4[k].z; v[3] += r4[k].w;
                    }
                    if (p.drop_thr) {
                        const uint64_t e0 = (uint64_t)orow[k] * (uint64_t)p.N + (uint64_t)col;
                        bool k0, k1, k2, k3;
                        chb_keep2((uint32_t)(e0 >> 1), p.drop_key, p.drop_thr, k0, k1);
                        chb_keep2((uint32_t)(e0 >> 1) + 1u, p.drop_key, p.drop_thr, k2, k3);
                        v[0] = k0 ? v[0] * p.drop_scale : 0.0f;
                        v[1] = k1 ? v[1] * p.drop_scale : 0.0f;
                        v[2] = k2 ? v[2] * p.drop_scale : 0.0f;
                        v[3] = k3 ? v[3] * p.drop_scale : 0.0f;
                    }
                    if (EPI == CHB_EPI_RESID) {
                        v[0] += r4[k].x; v[1] += r4[k].y; v[2] += r4[k].z; v[3] += r4[k].w;
                    }
                }
                char* dst = (EPI == CHB_EPI_PATCH) ? reinterpret_cast<char*>(p.C) + (orow[k] * p.ldc + col) * ESZ : cbase + rr * cstep;
                if (ok[k]) {
                    if (OUT == CHB_OUT_F32) {
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        uint2 o;
                        o.x = pack_bf16x2(v[0], v[1]);
                        o.y = pack_bf16x2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(dst) = o;
                    }
                }
            }
        }
    }
}

struct Cursor128 {
    int j, kt, m0, n0;
    bool valid;
};

__device__ __forceinline__ void cursor128_set(Cursor128& c, const TileWalk& w, int j) {
    c.j = j;
    c.kt = 0;
    const int li = w.slot + j * w.stride;
    c.valid = li < w.cnt;
    const int v = w.start + (c.valid ? li : 0);
    const int tm = v / w.tiles_n;
    c.m0 = tm * 128;
    c.n0 = (v - tm * w.tiles_n) * 256;
}
__device__ __forceinline__ void cursor128_next(Cursor128& c, const TileWalk& w) {
    if (++c.kt == w.ntk) cursor128_set(c, w, c.j + 1);
}

template <int EPI, int OUT>
__global__ void __launch_bounds__(256, 2) gemm_nt128_kernel(GemmParams p) {
    // [3 stages][A 128x32 | B 256x32] = 72 KiB + 4 x 2 KiB scratch = 80 KiB -> two workgroups per CU
    __shared__ __attribute__((aligned(16))) bf16_t smem[3 * 12288 + 4 * 1024];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int g = lane >> 4, i = lane & 15;

    TileWalk w;
    {
        const int nb = p.tiles_m * p.tiles_n;
        const int q = nb >> 3, r = nb & 7, x = blockIdx.x & 7;
        w.start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        w.cnt = q + (x < r ? 1 : 0);
        w.slot = blockIdx.x >> 3;
        w.stride = gridDim.x >> 3;
        w.tiles_n = p.tiles_n;
        w.ntk = p.K / 32;
        w.dyn = 0;
    }
    if (w.slot >= w.cnt) return;
    const int nmy = (w.cnt - w.slot + w.stride - 1) / w.stride;
    const int total = nmy * w.ntk;

    Cursor128 cs, cc;   // staging stream (2 steps ahead), compute position
    cursor128_set(cs, w, 0);
    cursor128_set(cc, w, 0);
    // prologue: stages 0 and 1
#pragma unroll
    for (int pre = 0; pre < 2; ++pre) {
        if (cs.valid) {
            bf16_t* slot = smem + pre * 12288;
            stage_k32(p.A, p.lda, cs.m0, p.M, cs.kt * 32, 8, slot, wave, lane);
            stage_k32(p.B, p.ldb, cs.n0, p.N, cs.kt * 32, 16, slot + 4096, wave, lane);
        }
        cursor128_next(cs, w);
    }

    float4_t acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};

    float* stage = reinterpret_cast<float*>(smem + 3 * 12288) + wave * 512;
    constexpr int NST = (EPI == CHB_EPI_GELU) ? 57 : 32;   // stores issued by one epilogue (capped so 6 + NST <= 63)
    int relax = 0;   // K-steps for which the just-issued epilogue stores may stay in flight

    for (int s = 0; s < total; ++s) {
        const int slot_id = s % 3;
        const bf16_t* As = smem + slot_id * 12288;
        const bf16_t* Bs = As + 4096;
        // stage s has landed once at most the next stage's 6 DMA instructions (and recent stores) are outstanding
        const bool next_in_flight = (s + 1 < total);
        if (next_in_flight) {
            if (relax > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (relax > 0) --relax;
        __builtin_amdgcn_s_barrier();
        // the slot read in step s-1 is free now: stage step s+2 into it
        if (cs.valid) {
            bf16_t* slot = smem + ((s + 2) % 3) * 12288;
            stage_k32(p.A, p.lda, cs.m0, p.M, cs.kt * 32, 8, slot, wave, lane);
            stage_k32(p.B, p.ldb, cs.n0, p.N, cs.kt * 32, 16, slot + 4096, wave, lane);
        }
        cursor128_next(cs, w);
        bf16x8_t af[8], bfr[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) bfr[b] = frag_k32(Bs, wave * 64 + b * 16 + i, g);
#pragma unroll
        for (int a = 0; a < 8; ++a) af[a] = frag_k32(As, a * 16 + i, g);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);

        if (cc.kt == w.ntk - 1) {
            if (cc.m0 + 128 <= p.M && cc.n0 + 256 <= p.N)
                epilogue_staged32<EPI, OUT, false>(p, stage, cc.m0, cc.n0 + wave * 64, acc, lane);
            else
                epilogue_staged32<EPI, OUT, true>(p, stage, cc.m0, cc.n0 + wave * 64, acc, lane);
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};
            relax = 2;   // the epilogue's loads were consumed (waited for) before its stores: only NST stores can be in flight
        }
        cursor128_next(cc, w);
    }
}

// ---- TN (wgrad), 256x256 output tile, one (tile, M-split) work item per 512-thread workgroup ----
// Four phases of 16 MFMAs over the same half-tile ring as gemm_nt256_kernel with TWO barriers per K-tile: after phase 2 (every
// wave has read this ring's dY slots, which phases 3/4 restage) and after the counted wait of phase 4 (step s+1 has landed; every
// wave has read this ring's X slots, which the next step restages).  Between them the waves drift by up to two phases, so one
// wave's transposed reads overlap another's MFMAs; since round 2 every wave also issues its own reads one half-phase ahead of the
// MFMAs that use them (inline asm, hand-counted waits - see the loop).  The operand half-tiles are [64 m][128 cols]
// (256-byte rows) and every fragment is a pair of transposed LDS reads.  The reduction axis is the
// long token axis M, split over workgroups so that tiles x splits ~ number of CUs; partial tiles are
// added into dW with fp32 atomics issued as full 256-byte rows (staged through LDS).
struct Tn256Params {
    const bf16_t* X; int64_t ldx;
    const bf16_t* Y; int64_t ldy;
    float* W; int64_t ldw;
    int M, Kd, Nd;
    int tiles_k, tiles_n, splits, steps_per_split;
    float* ws;        // optional [splits][Kd][Nd] fp32: every work item stores its partial tile there (plain 16-byte stores) and
                      // tn_reduce_kernel folds the planes into W; without it the partials meet in fp32 atomics on W
    float* colsum;    // COLSUM instantiation: fp32 [Nd] += column sums of dY (the bias gradient of the layer), see the kernel
};

__device__ __forceinline__ void stage_half_tn(const bf16_t* __restrict__ g, int64_t ld, int m0, int col0, int ncols, bf16_t* lds_half,
                                              int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int inst = j * 8 + wave;           // 16 instructions of 4 rows each
        const int r = inst * 4 + (lane >> 4);
        const int c = (lane & 15) ^ swz_tn(r);
        int gc = col0 + c * 8;
        gc = gc < ncols ? gc : ncols - 8;
        glds16(g + (int64_t)(m0 + r) * ld + gc, lds_half + inst * 512);
    }
}

// COLSUM: the column sums of dY over M (= ones^T . dY, the bias gradient) ride along as one more MFMA per dY fragment with an
// all-ones X fragment, in the waves of the first kd half (wk == 0) of the workgroups of the first kd tile (tk == 0): 8 MFMAs on
// top of 128 per SIMD and K-tile there, instead of a separate HBM pass over dY (465 MB for the QKV projection).  A separate
// instantiation, so the plain kernel keeps its registers and schedule.
// FAST (Kd % 256 == 0 and Nd % 256 == 0, no column clamp anywhere): the LDS-DMA addresses are a wave-uniform base (row of the K-step,
// column of the half-tile) + lane-constant 32-bit byte offsets instead of a 64-bit multiply-add per lane and instruction.
template <bool FAST>
__device__ __forceinline__ void stage_half_tn_t(const bf16_t* __restrict__ g, int64_t ld, int m0, int col0, int ncols, const uint32_t (&off)[2],
                                                bf16_t* lds_half, int wave, int lane) {
    if (FAST) {
        const char* base = reinterpret_cast<const char*>(g + (int64_t)m0 * ld + col0);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(reinterpret_cast<const bf16_t*>(base + off[j]), lds_half + (j * 8 + wave) * 512);
    } else {
        stage_half_tn(g, ld, m0, col0, ncols, lds_half, wave, lane);
    }
}

template <bool COLSUM, bool FAST>
__global__ void __launch_bounds__(512, 2) gemm_tn256_kernel(Tn256Params p) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 4 * 8192 + 8 * 2048];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wk = wave >> 2, wn = wave & 3;
    const int g = lane >> 4, i = lane & 15;
    const int tiles = p.tiles_k * p.tiles_n;
    // all output tiles of one M-split read the same X / dY rows: keep them on one XCD (one L2)
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int split = v / tiles;
    const int t = v - split * tiles;
    const int tk = t / p.tiles_n, tn = t - tk * p.tiles_n;
    const int k0 = tk * 256, n0 = tn * 256;
    const int total_steps = p.M / 64;
    const int s_begin = split * p.steps_per_split;
    int s_end = s_begin + p.steps_per_split;
    s_end = s_end < total_steps ? s_end : total_steps;
    const int total = s_end - s_begin;
    if (total <= 0) return;

    uint32_t offx[2] = {0, 0}, offy[2] = {0, 0};     // FAST: byte offsets of this lane's two LDS-DMA pieces (row 32 j + 4 wave + lane / 16)
    if (FAST) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (j * 8 + wave) * 4 + (lane >> 4);
            const int c = (lane & 15) ^ swz_tn(r);
            offx[j] = (uint32_t)(((int64_t)r * p.ldx + c * 8) * 2);
            offy[j] = (uint32_t)(((int64_t)r * p.ldy + c * 8) * 2);
        }
    }
    // prologue: all of step 0, dY halves of step 1
    stage_half_tn_t<FAST>(p.X, p.ldx, s_begin * 64, k0, p.Kd, offx, smem + 0 * 8192, wave, lane);
    stage_half_tn_t<FAST>(p.X, p.ldx, s_begin * 64, k0 + 128, p.Kd, offx, smem + 1 * 8192, wave, lane);
    stage_half_tn_t<FAST>(p.Y, p.ldy, s_begin * 64, n0, p.Nd, offy, smem + 2 * 8192, wave, lane);
    stage_half_tn_t<FAST>(p.Y, p.ldy, s_begin * 64, n0 + 128, p.Nd, offy, smem + 3 * 8192, wave, lane);
    if (total > 1) {
        stage_half_tn_t<FAST>(p.Y, p.ldy, (s_begin + 1) * 64, n0, p.Nd, offy, smem + (4 + 2) * 8192, wave, lane);
        stage_half_tn_t<FAST>(p.Y, p.ldy, (s_begin + 1) * 64, n0 + 128, p.Nd, offy, smem + (4 + 3) * 8192, wave, lane);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    float4_t acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (float4_t){0.f, 0.f, 0.f, 0.f};

    float4_t cacc[4];
    const bool do_colsum = COLSUM && tk == 0 && wk == 0;      // wave-uniform
    bf16x8_t ones;
    if (COLSUM) {
#pragma unroll
        for (int b = 0; b < 4; ++b) cacc[b] = (float4_t){0.f, 0.f, 0.f, 0.f};
        const short8_t o8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0
        ones = __builtin_bit_cast(bf16x8_t, o8);
    }
    const int ycol = (wn & 1) * 64;   // + nq*32 + b*16 within the wave's dY half
    // Lane-constant LDS element offsets of the transposed reads.  For row r0 = 8g + q the swizzle is the same at r0 + 4 and
    // r0 + 32, so one base per 16-column tile serves its four reads (rows +0, +4, +32, +36) with compile-time offsets.
    int xo[8], yo[4];
    {
        const int q = i >> 2, pp = i & 3;
        const int r0 = 8 * g + q;
        const int sw = swz_tn(r0);
#pragma unroll
        for (int a = 0; a < 8; ++a) xo[a] = r0 * 128 + ((((a * 16) >> 3) + (pp >> 1)) ^ sw) * 8 + 4 * (pp & 1);
#pragma unroll
        for (int b = 0; b < 4; ++b) yo[b] = r0 * 128 + ((((ycol + b * 16) >> 3) + (pp >> 1)) ^ sw) * 8 + 4 * (pp & 1);
    }

    // FAST: wave-uniform byte pointers of the two staging streams (X one K-step ahead, dY two), moved by one K-step of rows per
    // iteration - a piece is then one 64-bit add in front of the LDS-DMA, not a 64-bit multiply-add chain per half-tile
    const int64_t xstep = (int64_t)64 * p.ldx * 2, ystep = (int64_t)64 * p.ldy * 2;
    const char* xnext = reinterpret_cast<const char*>(p.X + (int64_t)(s_begin + 1) * 64 * p.ldx + k0);
    const char* ynext = reinterpret_cast<const char*>(p.Y + (int64_t)(s_begin + 2) * 64 * p.ldy + n0);
    auto stage_fast = [&](const char* base, const uint32_t (&off)[2], bf16_t* lds_half) {
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(reinterpret_cast<const bf16_t*>(base + off[j]), lds_half + (j * 8 + wave) * 512);
    };
    // Fragment reads run AHEAD of the MFMAs that use them (tools ablation: without any reads the kernel is 27 % faster, the reads
    // were a burst in front of every 16-MFMA block): inline asm transposed reads with hand-counted lgkmcnt, as in
    // gemm_nt256sp_kernel (hipcc turns every LDS wait into lgkmcnt(0) while an LDS-DMA is in flight).  Same 64 fragment registers:
    //   phase 1 issues y1 (phase 2's dY columns);  phase 2 re-loads each k-half of xf with kd rows 64-127 as soon as its MFMAs on
    //   that half are issued (phase 3's X);  phase 4 does the same with the NEXT step's kd rows 0-63 and dY columns 0-31.
    // Same MFMA order per accumulator as before.
    uint32_t xadr[8], yadr[4];
#pragma unroll
    for (int a = 0; a < 8; ++a) xadr[a] = lds_addr(smem + wk * 8192 + xo[a]);
#pragma unroll
    for (int b = 0; b < 4; ++b) yadr[b] = lds_addr(smem + (2 + (wn >> 1)) * 8192 + yo[b]);
    bf16x8_t xf[4][2], y0[2][2], y1[2][2];
#define TN_WAIT_X(n, k) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(xf[0][k]), "+v"(xf[1][k]), "+v"(xf[2][k]), "+v"(xf[3][k]));
#define TN_WAIT_Y(n, y) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(y[0][0]), "+v"(y[0][1]), "+v"(y[1][0]), "+v"(y[1][1]));
    // first step: kd rows 0-63 and dY columns 0-31 of ring 0 (landed: the prologue's barrier)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int a = 0; a < 4; ++a) { if (ks == 0) tr_read128<0>(xf[a][0], xadr[a]); else tr_read128<8192>(xf[a][1], xadr[a]); }
#pragma unroll
        for (int b = 0; b < 2; ++b) { if (ks == 0) tr_read128<0>(y0[b][0], yadr[b]); else tr_read128<8192>(y0[b][1], yadr[b]); }
    }
    for (int s = 0; s < total; ++s) {
        bf16_t* ring = smem + (s & 1) * 4 * 8192;
        bf16_t* nring = smem + ((s + 1) & 1) * 4 * 8192;
        const uint32_t ro = (s & 1) * 65536, nro = 65536 - ro;
        const bool have1 = s + 1 < total, have2 = s + 2 < total;
        const int m1 = (s_begin + s + 1) * 64, m2 = (s_begin + s + 2) * 64;

        // phase 1: (kd 0-63, nd 0-31).  In flight on entry: xf (16 reads), y0 (8 reads).
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            tr_read128<0>(y1[b][0], yadr[2 + b] + ro);
            tr_read128<8192>(y1[b][1], yadr[2 + b] + ro);
        }
        if (have1) {
            if (FAST) stage_fast(xnext, offx, nring + 0 * 8192);
            else stage_half_tn(p.X, p.ldx, m1, k0, p.Kd, nring + 0 * 8192, wave, lane);
        }
        TN_WAIT_X(8, 0) TN_WAIT_X(8, 1) TN_WAIT_Y(8, y0)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y0[b][ks], xf[a][ks], acc[a][b], 0, 0, 0);
        if (COLSUM && do_colsum) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int b = 0; b < 2; ++b) cacc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y0[b][ks], ones, cacc[b], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);

        // phase 2: (kd 0-63, nd 32-63); each k-half of xf is re-loaded with kd rows 64-127 behind its MFMAs
        if (have1) {
            if (FAST) stage_fast(xnext + 256, offx, nring + 1 * 8192);
            else stage_half_tn(p.X, p.ldx, m1, k0 + 128, p.Kd, nring + 1 * 8192, wave, lane);
        }
        TN_WAIT_Y(0, y1)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][2 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y1[b][ks], xf[a][ks], acc[a][2 + b], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 4; ++a) { if (ks == 0) tr_read128<0>(xf[a][0], xadr[4 + a] + ro); else tr_read128<8192>(xf[a][1], xadr[4 + a] + ro); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (COLSUM && do_colsum) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int b = 0; b < 2; ++b) cacc[2 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y1[b][ks], ones, cacc[2 + b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();        // every wave's dY reads of this ring have returned (waited for in phases 1 and 2)

        // phase 3: (kd 64-127, nd 32-63); the dY slots of this ring are free now
        if (have2) {
            if (FAST) stage_fast(ynext, offy, ring + 2 * 8192);
            else stage_half_tn(p.Y, p.ldy, m2, n0, p.Nd, ring + 2 * 8192, wave, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 0) { TN_WAIT_X(8, 0) } else { TN_WAIT_X(0, 1) }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[4 + a][2 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y1[b][ks], xf[a][ks], acc[4 + a][2 + b], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // phase 4: (kd 64-127, nd 0-31); retire step s+1's loads, then re-load xf / y0 with the next step's first fragments
        if (have2) {
            if (FAST) stage_fast(ynext + 256, offy, ring + 3 * 8192);
            else stage_half_tn(p.Y, p.ldy, m2, n0 + 128, p.Nd, ring + 3 * 8192, wave, lane);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();        // step s+1 has landed; every wave's X reads of this ring have returned (phase 3's waits)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[4 + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y0[b][ks], xf[a][ks], acc[4 + a][b], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (have1) {
#pragma unroll
                for (int a = 0; a < 4; ++a) { if (ks == 0) tr_read128<0>(xf[a][0], xadr[a] + nro); else tr_read128<8192>(xf[a][1], xadr[a] + nro); }
#pragma unroll
                for (int b = 0; b < 2; ++b) { if (ks == 0) tr_read128<0>(y0[b][0], yadr[b] + nro); else tr_read128<8192>(y0[b][1], yadr[b] + nro); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        xnext += xstep;
        ynext += ystep;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef TN_WAIT_X
#undef TN_WAIT_Y

    // acc[a][b][r]: kd = k0 + wk*128 + a*16 + i, nd = n0 + wn*64 + b*16 + 4g + r.  Stage one 16(kd) x 64(nd) tile row
    // at a time and add it as 16 full 256-byte rows.
    if (COLSUM && do_colsum) {
        // cacc[b][r] = sum over this split's rows of dY[:, n0 + wn*64 + b*16 + 4g + r], replicated over the 16 lanes i
        if (i == 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ndc = n0 + wn * 64 + b * 16 + 4 * g + r;
                    if (ndc < p.Nd) atomicAdd(p.colsum + ndc, cacc[b][r]);
                }
        }
    }
    float* stage = reinterpret_cast<float*>(smem + 2 * 4 * 8192) + wave * 1024;
    if (p.ws) {
        // partial plane of this split: lane = (row r4 + 4k of the tile row, 4 columns) -> 4 rows x 256 bytes per store instruction
        float* plane = p.ws + (int64_t)split * p.Kd * p.Nd;
        const int r4 = lane >> 4, c4 = lane & 15;
        const int nd4 = n0 + wn * 64 + 4 * c4;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
#pragma unroll
            for (int b = 0; b < 4; ++b) *reinterpret_cast<float4_t*>(stage + i * 64 + (((4 * b + g) ^ i) << 2)) = acc[a][b];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * k + r4;
                const float4_t v = *reinterpret_cast<const float4_t*>(stage + r * 64 + ((c4 ^ r) << 2));
                const int kd = k0 + wk * 128 + a * 16 + r;
                if (kd < p.Kd && nd4 < p.Nd) *reinterpret_cast<float4_t*>(plane + (int64_t)kd * p.Nd + nd4) = v;
            }
        }
        return;
    }
    const int nd = n0 + wn * 64 + lane;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) *reinterpret_cast<float4_t*>(stage + i * 64 + (((4 * b + g) ^ i) << 2)) = acc[a][b];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = stage[r * 64 + ((((lane >> 2) ^ r) << 2) | (lane & 3))];
            const int kd = k0 + wk * 128 + a * 16 + r;
            if (kd < p.Kd && nd < p.Nd) atomicAdd(p.W + (int64_t)kd * p.ldw + nd, v);
        }
    }
}

// W[kd][nd] += sum over the splits of ws[s][kd][nd]  (one thread = 4 columns; Nd % 4 == 0)
__global__ void __launch_bounds__(256) tn_reduce_kernel(const float* __restrict__ ws, int splits, int Kd, int Nd, float* __restrict__ W,
                                                        int64_t ldw) {
    const int nq = Nd >> 2;
    const int64_t total = (int64_t)Kd * nq, plane = (int64_t)Kd * Nd;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int kd = (int)(idx / nq), c = (int)(idx - (int64_t)kd * nq) * 4;
        const float* src = ws + (int64_t)kd * Nd + c;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sp = 0; sp < splits; ++sp) {
            const float4 v = *reinterpret_cast<const float4*>(src + sp * plane);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        float4* dst = reinterpret_cast<float4*>(W + (int64_t)kd * ldw + c);
        float4 w = *dst;
        w.x += acc.x; w.y += acc.y; w.z += acc.z; w.w += acc.w;
        *dst = w;
    }
}

// the same for up to four gradients in ONE launch (a block's four weight-gradient GEMMs): block b belongs to the item whose block
// range holds it; per item exactly tn_reduce_kernel's arithmetic (same summation order: bit-identical to four launches)
struct FoldItem { const float* ws; float* W; int64_t ldw; int splits, Kd, Nd, first_block, blocks; };
struct FoldBatch { FoldItem it[4]; int n; };
__global__ void __launch_bounds__(256) tn_reduce_multi_kernel(FoldBatch fb) {
    int k = 0;
    while (k + 1 < fb.n && (int)blockIdx.x >= fb.it[k + 1].first_block) ++k;
    const FoldItem f = fb.it[k];
    const int nq = f.Nd >> 2;
    const int64_t total = (int64_t)f.Kd * nq, plane = (int64_t)f.Kd * f.Nd;
    for (int64_t idx = (int64_t)((int)blockIdx.x - f.first_block) * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)f.blocks * blockDim.x) {
        const int kd = (int)(idx / nq), c = (int)(idx - (int64_t)kd * nq) * 4;
        const float* src = f.ws + (int64_t)kd * f.Nd + c;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sp = 0; sp < f.splits; ++sp) {
            const float4 v = *reinterpret_cast<const float4*>(src + sp * plane);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        float4* dst = reinterpret_cast<float4*>(f.W + (int64_t)kd * f.ldw + c);
        float4 w = *dst;
        w.x += acc.x; w.y += acc.y; w.z += acc.z; w.w += acc.w;
        *dst = w;
    }
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// 0 = automatic, 1 = 128x128 tiles (one workgroup per tile), 2 = persistent 256x256 tiles (lockstep), 3 = persistent 128x256 x 2 WG/CU,
// 4 = persistent 256x256 ping-pong, 5 = persistent 256x256 pipelined reads + spread staging (the automatic choice for large shapes)
int gemm_algo_override() { return chb_option(CHB_OPT_GEMM_ALGO); }

template <int EPI>
int launch_nt(GemmParams p, int out_dtype, hipStream_t s) {
    int algo = gemm_algo_override();
    // automatic: the persistent 256x256 kernel with pipelined reads and spread staging where every tile is full (both persistent kernels
    // have the tile queue)
    const bool full_tiles = !(p.M & 255) && !(p.N & 255) && (out_dtype == CHB_OUT_F32 || (!(p.ldc & 7) && !(p.aux && (p.ld_aux & 7))));
    if (algo == 0) algo = (p.M >= 2048 && p.N >= 256) ? (full_tiles ? 5 : 2) : 1;      // ragged shapes: the lockstep kernel is the faster one
    if (p.colsum && algo != 2 && algo != 4 && algo != 5) {
        // only the persistent 256x256 kernel fuses the column sums; other paths add them with the stand-alone pass
        if (out_dtype != CHB_OUT_BF16) return CHB_EUNSUPPORTED;
        float* cs = p.colsum;
        p.colsum = nullptr;
        const int rc = launch_nt<EPI>(p, out_dtype, s);
        if (rc != CHB_OK) return rc;
        return chb_colsum_bf16(p.C, p.ldc, cs, p.M, p.N, (void*)s);
    }
    if (algo == 3) {
        p.tiles_m = chb_div_up(p.M, 128);
        p.tiles_n = chb_div_up(p.N, 256);
        int grid = (2 * num_cus()) & ~7;
        if (grid < 8) grid = 8;
        const dim3 g(grid), block(256);
        if (out_dtype == CHB_OUT_F32) hipLaunchKernelGGL((gemm_nt128_kernel<EPI, CHB_OUT_F32>), g, block, 0, s, p);
        else hipLaunchKernelGGL((gemm_nt128_kernel<EPI, CHB_OUT_BF16>), g, block, 0, s, p);
        return CHB_OK;
    }
    if (algo == 4) {     // ping-pong schedule: full tiles and >= 4 K-tiles only
        if ((p.M & 255) || (p.N & 255) || p.K < 4 * BK || (p.ldc & 7) || (p.aux && (p.ld_aux & 7))) algo = 2;
    }
    if (algo == 4) {
        p.tiles_m = p.M / 256;
        p.tiles_n = p.N / 256;
        int grid = num_cus() & ~7;
        if (grid < 8) grid = 8;
        const dim3 g(grid), block(512);
        if (out_dtype == CHB_OUT_F32) hipLaunchKernelGGL((gemm_nt256pp_kernel<EPI, CHB_OUT_F32>), g, block, 0, s, p);
        else hipLaunchKernelGGL((gemm_nt256pp_kernel<EPI, CHB_OUT_BF16>), g, block, 0, s, p);
        return CHB_OK;
    }
    if (algo == 5) {     // software-pipelined fragment reads, staging spread over the K-step
        p.tiles_m = chb_div_up(p.M, 256);
        p.tiles_n = chb_div_up(p.N, 256);
        int grid = num_cus() & ~7;
        if (grid < 8) grid = 8;
        if (chb_option(CHB_OPT_DEBUG) >= 8) grid = chb_option(CHB_OPT_DEBUG) & ~7;   // timing experiments: fewer workgroups than CUs
        const dim3 g(grid), block(512);
        const bool fast = !(p.M & 255) && !(p.N & 255) && (out_dtype == CHB_OUT_F32 || (!(p.ldc & 7) && !(p.aux && (p.ld_aux & 7))));
        if (out_dtype == CHB_OUT_F32) {
            if (fast) hipLaunchKernelGGL((gemm_nt256sp_kernel<EPI, CHB_OUT_F32, true>), g, block, 0, s, p);
            else hipLaunchKernelGGL((gemm_nt256sp_kernel<EPI, CHB_OUT_F32, false>), g, block, 0, s, p);
        } else {
            if (fast) hipLaunchKernelGGL((gemm_nt256sp_kernel<EPI, CHB_OUT_BF16, true>), g, block, 0, s, p);
            else hipLaunchKernelGGL((gemm_nt256sp_kernel<EPI, CHB_OUT_BF16, false>), g, block, 0, s, p);
        }
        return CHB_OK;
    }
    if (algo == 2) {
        p.tiles_m = chb_div_up(p.M, 256);
        p.tiles_n = chb_div_up(p.N, 256);
        int grid = num_cus() & ~7;
        if (grid < 8) grid = 8;
        const dim3 g(grid), block(512);
        // every tile full: the clamped staging and the guarded epilogue are compiled out; bf16 rows leave as 16-byte stores (WIDE)
        const bool fast = !(p.M & 255) && !(p.N & 255) && (out_dtype == CHB_OUT_F32 || (!(p.ldc & 7) && !(p.aux && (p.ld_aux & 7))));
        if (out_dtype == CHB_OUT_F32) {
            if (fast) hipLaunchKernelGGL((gemm_nt256_kernel<EPI, CHB_OUT_F32, true>), g, block, 0, s, p);
            else hipLaunchKernelGGL((gemm_nt256_kernel<EPI, CHB_OUT_F32, false>), g, block, 0, s, p);
        } else {
            if (fast) hipLaunchKernelGGL((gemm_nt256_kernel<EPI, CHB_OUT_BF16, true>), g, block, 0, s, p);
            else hipLaunchKernelGGL((gemm_nt256_kernel<EPI, CHB_OUT_BF16, false>), g, block, 0, s, p);
        }
        return CHB_OK;
    }
    const dim3 grid(p.tiles_m * p.tiles_n), block(256);
    if (out_dtype == CHB_OUT_F32) hipLaunchKernelGGL((gemm_nt_kernel<EPI, CHB_OUT_F32>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<EPI, CHB_OUT_BF16>), grid, block, 0, s, p);
    return CHB_OK;
}

}  // namespace

extern "C" {

int chb_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                const float* bias, int epilogue, int out_dtype, void* aux, int64_t ld_aux, const float* resid,
                int64_t ld_resid, int period, float drop_rate, uint32_t drop_key, float* out_colsum, void* stream) {
    if (!A || !B || !C || M < 0 || N <= 0 || K <= 0) return CHB_EINVAL;
    if (M == 0) return CHB_OK;
    if (K % BK != 0 || (N & 3) || (lda & 7) || (ldb & 7) || (ldc & 3)) return CHB_EUNSUPPORTED;
    if (out_dtype != CHB_OUT_BF16 && out_dtype != CHB_OUT_F32) return CHB_EINVAL;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15)) return CHB_EINVAL;
    if (out_colsum && epilogue == CHB_EPI_PATCH) return CHB_EUNSUPPORTED;
    if ((epilogue == CHB_EPI_GELU || epilogue == CHB_EPI_DGELU) && (!aux || (ld_aux & 3))) return CHB_EINVAL;
    if ((epilogue == CHB_EPI_RESID || epilogue == CHB_EPI_PATCH) && (!resid || (ld_resid & 3))) return CHB_EINVAL;
    if (epilogue == CHB_EPI_PATCH && period <= 0) return CHB_EINVAL;
    if (drop_rate < 0.0f || drop_rate >= 1.0f) return CHB_EINVAL;
    GemmParams p;
    p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.bias = bias; p.aux = (bf16_t*)aux; p.ld_aux = ld_aux;
    p.resid = resid; p.ld_resid = ld_resid;
    p.period = period & 0xffffff;             // bits 0..23: patches per image
    p.n_special = 1 + ((period >> 24) & 15);  // bits 24..27: special tokens beyond the class token
    p.drop_thr = drop_rate > 0.0f ? chb_drop_threshold(drop_rate) : 0u;
    p.drop_scale = 1.0f / (1.0f - drop_rate);
    p.drop_key = drop_key;
    p.tiles_m = chb_div_up(M, BM); p.tiles_n = chb_div_up(N, BN);
    p.colsum = out_colsum;
    p.walk_panel = chb_option(CHB_OPT_GEMM_WALK);   // 0 = linear tile ids per XCD, 1 (default) = panel walk where it pays, 2 = always panel
    {   // tile queue of the persistent kernel: a fresh slot per launch (launches in flight at once never share one)
        static std::atomic<unsigned> launch_seq{0};
        p.queue_slot = chb_option(CHB_OPT_GEMM_TILE_QUEUE) ? (int)(launch_seq.fetch_add(1, std::memory_order_relaxed) % TILE_QUEUE_SLOTS) : -1;
    }
    hipStream_t s = (hipStream_t)stream;
    const int prof = chb_prof_begin(0, epilogue, out_dtype, M, N, K, s);
    switch (epilogue) {
        int rc;
        case CHB_EPI_NONE: rc = launch_nt<CHB_EPI_NONE>(p, out_dtype, s); if (rc) return rc; break;
        case CHB_EPI_GELU: rc = launch_nt<CHB_EPI_GELU>(p, out_dtype, s); if (rc) return rc; break;
        case CHB_EPI_DGELU: rc = launch_nt<CHB_EPI_DGELU>(p, out_dtype, s); if (rc) return rc; break;
        case CHB_EPI_RESID: rc = launch_nt<CHB_EPI_RESID>(p, out_dtype, s); if (rc) return rc; break;
        case CHB_EPI_PATCH: rc = launch_nt<CHB_EPI_PATCH>(p, out_dtype, s); if (rc) return rc; break;
        default: return CHB_EINVAL;
    }
    if (prof >= 0) {      // which kernel launch_nt picked (same rule)
        int algo = gemm_algo_override();
        const bool full_tiles = !(M & 255) && !(N & 255) && (out_dtype == CHB_OUT_F32 || (!(ldc & 7) && !(aux && (ld_aux & 7))));
        if (algo == 0) algo = (M >= 2048 && N >= 256) ? (full_tiles ? 5 : 2) : 1;
        if (algo == 4 && ((M & 255) || (N & 255) || K < 4 * BK || (ldc & 7) || (aux && (ld_aux & 7)))) algo = 2;
        chb_prof_end(prof, algo, s);
    }
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_gemm_tn(const void* X, int64_t ldx, const void* dY, int64_t ldy, float* dW, int64_t ldw, int M, int Kd, int Nd,
                void* stream) {
    return chb_gemm_tn_ws(X, ldx, dY, ldy, dW, ldw, M, Kd, Nd, nullptr, 0, 1, nullptr, stream);
}

// split plan of the persistent 256x256 weight-gradient kernel; `planes`: partial planes in caller scratch instead of atomics
struct Tn256Plan { bool use256; int tiles_k, tiles_n, splits, steps_per_split; bool planes; };

static Tn256Plan tn256_plan(int M, int Kd, int Nd, const float* workspace, int64_t workspace_bytes, const float* dW, int64_t ldw) {
    Tn256Plan pl;
    int algo = gemm_algo_override();
    if (algo == 0) algo = (M >= 4096 && Kd >= 128 && Nd >= 128) ? 2 : 1;
    pl.use256 = algo == 2;
    pl.tiles_k = chb_div_up(Kd, 256); pl.tiles_n = chb_div_up(Nd, 256);
    const int tiles = pl.tiles_k * pl.tiles_n;
    const int steps = M / 64;
    int splits = num_cus() / tiles;          // one work item per CU
    if (splits < 1) splits = 1;
    if (splits > steps) splits = steps;
    pl.steps_per_split = chb_div_up(steps, splits);
    pl.splits = chb_div_up(steps, pl.steps_per_split);
    // partial planes + one fold launch when the caller lends enough scratch (and the fold's float4 accesses are aligned)
    const int64_t need = (int64_t)pl.splits * Kd * Nd * 4;
    const bool force_atomics = chb_option(CHB_OPT_TN_ATOMICS) == 1;     // A/B timing
    pl.planes = pl.use256 && workspace && workspace_bytes >= need && pl.splits > 1 && !(Nd & 3) && !(ldw & 3) &&
                !((uintptr_t)workspace & 15) && !((uintptr_t)dW & 15) && !force_atomics;
    return pl;
}

int chb_gemm_tn_fold(const float* workspace, int64_t workspace_bytes, float* dW, int64_t ldw, int M, int Kd, int Nd, void* stream) {
    if (!dW || M < 0 || Kd <= 0 || Nd <= 0) return CHB_EINVAL;
    if (M == 0 || M % 64 != 0) return M == 0 ? CHB_OK : CHB_EUNSUPPORTED;
    const Tn256Plan pl = tn256_plan(M, Kd, Nd, workspace, workspace_bytes, dW, ldw);
    if (!pl.planes) return CHB_OK;             // the GEMM took the atomic epilogue: nothing to fold
    const int64_t quads = (int64_t)Kd * (Nd / 4);
    const int64_t blocks = (quads + 255) / 256;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, workspace,
                       pl.splits, Kd, Nd, dW, ldw);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_gemm_tn_fold_multi(const chb_tn_fold_item* items_host, int n_items, void* stream) {
    if (n_items < 0 || n_items > 4 || (n_items && !items_host)) return CHB_EINVAL;
    FoldBatch fb;
    fb.n = 0;
    int at = 0;
    for (int k = 0; k < n_items; ++k) {
        const chb_tn_fold_item& it = items_host[k];
        if (!it.dW || it.M < 0 || it.Kd <= 0 || it.Nd <= 0) return CHB_EINVAL;
        if (it.M == 0) continue;
        if (it.M % 64 != 0) return CHB_EUNSUPPORTED;
        const Tn256Plan pl = tn256_plan(it.M, it.Kd, it.Nd, it.workspace, it.workspace_bytes, it.dW, it.ldw);
        if (!pl.planes) continue;              // that GEMM took the atomic epilogue: nothing to fold
        const int64_t quads = (int64_t)it.Kd * (it.Nd / 4);
        const int64_t blocks = (quads + 255) / 256;
        FoldItem& f = fb.it[fb.n++];
        f.ws = it.workspace; f.W = it.dW; f.ldw = it.ldw; f.splits = pl.splits; f.Kd = it.Kd; f.Nd = it.Nd;
        f.first_block = at;
        f.blocks = (int)(blocks < 8192 ? blocks : 8192);
        at += f.blocks;
    }
    if (!fb.n) return CHB_OK;
    hipLaunchKernelGGL(tn_reduce_multi_kernel, dim3((unsigned)at), dim3(256), 0, (hipStream_t)stream, fb);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_gemm_tn_ws(const void* X, int64_t ldx, const void* dY, int64_t ldy, float* dW, int64_t ldw, int M, int Kd, int Nd,
                   float* workspace, int64_t workspace_bytes, int fold, float* dy_colsum, void* stream) {
    if (!X || !dY || !dW || M < 0 || Kd <= 0 || Nd <= 0) return CHB_EINVAL;
    if (M == 0) return CHB_OK;
    if (M % 64 != 0 || (Kd & 7) || (Nd & 7) || (ldx & 7) || (ldy & 7)) return CHB_EUNSUPPORTED;
    if (((uintptr_t)X & 15) || ((uintptr_t)dY & 15)) return CHB_EINVAL;
    {
        const Tn256Plan pl = tn256_plan(M, Kd, Nd, workspace, workspace_bytes, dW, ldw);
        if (pl.use256) {
            Tn256Params q;
            q.X = (const bf16_t*)X; q.ldx = ldx; q.Y = (const bf16_t*)dY; q.ldy = ldy; q.W = dW; q.ldw = ldw;
            q.M = M; q.Kd = Kd; q.Nd = Nd;
            q.tiles_k = pl.tiles_k; q.tiles_n = pl.tiles_n;
            q.steps_per_split = pl.steps_per_split;
            q.splits = pl.splits;
            q.ws = pl.planes ? workspace : nullptr;
            q.colsum = dy_colsum;
            const dim3 grid(pl.tiles_k * pl.tiles_n * q.splits);
            const bool fast = !(Kd & 255) && !(Nd & 255) && chb_option(CHB_OPT_TN_FAST) != 0;   // 0 = generic staging addresses (A/B timing)
            hipStream_t st = (hipStream_t)stream;
            const int prof = chb_prof_begin(1, dy_colsum ? 1 : 0, CHB_OUT_F32, Kd, Nd, M, st);
            if (dy_colsum) {
                if (fast) hipLaunchKernelGGL((gemm_tn256_kernel<true, true>), grid, dim3(512), 0, st, q);
                else hipLaunchKernelGGL((gemm_tn256_kernel<true, false>), grid, dim3(512), 0, st, q);
            } else {
                if (fast) hipLaunchKernelGGL((gemm_tn256_kernel<false, true>), grid, dim3(512), 0, st, q);
                else hipLaunchKernelGGL((gemm_tn256_kernel<false, false>), grid, dim3(512), 0, st, q);
            }
            chb_prof_end(prof, 1, st);
            CHB_LAUNCH_CHECK();
            if (pl.planes && fold) return chb_gemm_tn_fold(workspace, workspace_bytes, dW, ldw, M, Kd, Nd, stream);
            return CHB_OK;
        }
    }
    TnParams p;
    p.X = (const bf16_t*)X; p.ldx = ldx; p.Y = (const bf16_t*)dY; p.ldy = ldy; p.W = dW; p.ldw = ldw;
    p.M = M; p.Kd = Kd; p.Nd = Nd;
    p.tiles_k = chb_div_up(Kd, 128); p.tiles_n = chb_div_up(Nd, 128);
    const int tiles = p.tiles_k * p.tiles_n;
    const int steps = M / 64;
    int splits = chb_div_up(1536, tiles);  // ~6 workgroups per CU in flight
    if (splits > steps) splits = steps;
    if (splits < 1) splits = 1;
    p.steps_per_split = chb_div_up(steps, splits);
    p.splits = chb_div_up(steps, p.steps_per_split);
    const int prof = chb_prof_begin(1, dy_colsum ? 1 : 0, CHB_OUT_F32, Kd, Nd, M, (hipStream_t)stream);
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * p.splits), dim3(256), 0, (hipStream_t)stream, p);
    chb_prof_end(prof, 0, (hipStream_t)stream);
    CHB_LAUNCH_CHECK();
    if (dy_colsum) return chb_colsum_bf16(dY, ldy, dy_colsum, M, Nd, stream);     // small shapes: the stand-alone pass
    return CHB_OK;
}

// The tile queue's per-launch-slot, per-XCD counters (g_tile_ctr) are the one piece of device state this library keeps between
// launches.  A launch leaves its slot at zero (the holder of an XCD's last ticket clears it), so nothing needs doing in normal
// operation; a launch that FAULTED or was aborted mid-way can leave a slot dirty, and the next launch on that slot would then
// skip tiles.  This entry zeroes all counters on `stream` (ordered behind whatever runs there): the engine calls it when it is
// constructed, and a caller that catches a device error calls it before re-using the library with GEMM_TILE_QUEUE on.
int chb_gemm_tile_queue_reset(void* stream) {
    void* dev = nullptr;
    if (hipGetSymbolAddress(&dev, HIP_SYMBOL(g_tile_ctr)) != hipSuccess) return CHB_ELAUNCH;
    if (hipMemsetAsync(dev, 0, sizeof(int) * TILE_QUEUE_SLOTS * 8, (hipStream_t)stream) != hipSuccess) return CHB_ELAUNCH;
    return CHB_OK;
}

#ifdef CHB_PHASE_STAMPS
int chb_debug_phase_stamps(unsigned int* host_out, int n_groups) {   // diagnostic build only
    if (n_groups > 1024) n_groups = 1024;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_stamps), sizeof(unsigned int) * 20 * n_groups) == hipSuccess ? 0 : -1;
}
#endif
#ifdef CHB_CLOCK_STAMPS
int chb_debug_clock_stamps(unsigned long long* host_out, int n_groups) {   // diagnostic build only, not part of the ABI
    if (n_groups > 1024) n_groups = 1024;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clock_stamps), sizeof(unsigned long long) * 4 * n_groups) == hipSuccess ? 0 : -1;
}
#endif
}  // extern "C"
