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

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_ELEMS = 128 * 64;  // one operand tile, either orientation

struct GemmParams {
    const bf16_t* A; int64_t lda;
    const bf16_t* B; int64_t ldb;
    void* C; int64_t ldc;
    int M, N, K;
    const float* bias;
    bf16_t* aux; int64_t ld_aux;
    const float* resid; int64_t ld_resid;
    int period;
    float drop_scale; uint32_t drop_thr; uint32_t drop_key;
    int tiles_m, tiles_n;
};

// bijective XCD-aware remap: consecutive virtual ids (which share an A panel) stay on one XCD
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int x = bid & 7, idx = bid >> 3;
    const int start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return start + idx;
}

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

template <int EPI, int OUT>
__device__ __forceinline__ void epilogue4(const GemmParams& p, int row, int col, float4_t acc) {
    // row < M, col % 4 == 0, col < N
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    int64_t orow = row;
    if (EPI == CHB_EPI_GELU) {
        uint2 a;
        a.x = pack_bf16x2(v[0], v[1]);
        a.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p.aux + (int64_t)row * p.ld_aux + col) = a;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gelu_f(v[i]);
    } else if (EPI == CHB_EPI_DGELU) {
        const uint2 a = *reinterpret_cast<const uint2*>(p.aux + (int64_t)row * p.ld_aux + col);
        v[0] *= dgelu_f(bf16_to_f32((bf16_t)(a.x & 0xffff)));
        v[1] *= dgelu_f(bf16_to_f32((bf16_t)(a.x >> 16)));
        v[2] *= dgelu_f(bf16_to_f32((bf16_t)(a.y & 0xffff)));
        v[3] *= dgelu_f(bf16_to_f32((bf16_t)(a.y >> 16)));
    } else if (EPI == CHB_EPI_RESID || EPI == CHB_EPI_PATCH) {
        const float* rs;
        if (EPI == CHB_EPI_PATCH) {
            const int b = row / p.period, pp = row - b * p.period;
            orow = (int64_t)b * (p.period + 1) + 1 + pp;
            rs = p.resid + (int64_t)(1 + pp) * p.ld_resid + col;
            const float4 r4 = *reinterpret_cast<const float4*>(rs);
            v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;  // + positional embedding, then dropout
        }
        if (p.drop_thr) {
            const uint64_t e0 = (uint64_t)orow * (uint64_t)p.N + (uint64_t)col;  // even (N % 4 == 0)
            bool k0, k1, k2, k3;
            chb_keep2((uint32_t)(e0 >> 1), p.drop_key, p.drop_thr, k0, k1);
            chb_keep2((uint32_t)(e0 >> 1) + 1u, p.drop_key, p.drop_thr, k2, k3);
            v[0] = k0 ? v[0] * p.drop_scale : 0.0f;
            v[1] = k1 ? v[1] * p.drop_scale : 0.0f;
            v[2] = k2 ? v[2] * p.drop_scale : 0.0f;
            v[3] = k3 ? v[3] * p.drop_scale : 0.0f;
        }
        if (EPI == CHB_EPI_RESID) {
            const float4 r4 = *reinterpret_cast<const float4*>(p.resid + (int64_t)row * p.ld_resid + col);
            v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
    }
    if (OUT == CHB_OUT_F32) {
        *reinterpret_cast<float4*>((float*)p.C + orow * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>((bf16_t*)p.C + orow * p.ldc + col) = o;
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

#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int row = m0 + wm * 64 + a * 16 + i;
        if (row < p.M) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int col = n0 + wn * 64 + b * 16 + g * 4;
                if (col < p.N) epilogue4<EPI, OUT>(p, row, col, acc[a][b]);
            }
        }
    }
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

template <int EPI>
int launch_nt(const GemmParams& p, int out_dtype, hipStream_t s) {
    const dim3 grid(p.tiles_m * p.tiles_n), block(256);
    if (out_dtype == CHB_OUT_F32) hipLaunchKernelGGL((gemm_nt_kernel<EPI, CHB_OUT_F32>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<EPI, CHB_OUT_BF16>), grid, block, 0, s, p);
    return CHB_OK;
}

}  // namespace

extern "C" {

int chb_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                const float* bias, int epilogue, int out_dtype, void* aux, int64_t ld_aux, const float* resid,
                int64_t ld_resid, int period, float drop_rate, uint32_t drop_key, void* stream) {
    if (!A || !B || !C || M < 0 || N <= 0 || K <= 0) return CHB_EINVAL;
    if (M == 0) return CHB_OK;
    if (K % BK != 0 || (N & 3) || (lda & 7) || (ldb & 7) || (ldc & 3)) return CHB_EUNSUPPORTED;
    if (out_dtype != CHB_OUT_BF16 && out_dtype != CHB_OUT_F32) return CHB_EINVAL;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15)) return CHB_EINVAL;
    if ((epilogue == CHB_EPI_GELU || epilogue == CHB_EPI_DGELU) && (!aux || (ld_aux & 3))) return CHB_EINVAL;
    if ((epilogue == CHB_EPI_RESID || epilogue == CHB_EPI_PATCH) && (!resid || (ld_resid & 3))) return CHB_EINVAL;
    if (epilogue == CHB_EPI_PATCH && period <= 0) return CHB_EINVAL;
    if (drop_rate < 0.0f || drop_rate >= 1.0f) return CHB_EINVAL;
    GemmParams p;
    p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.bias = bias; p.aux = (bf16_t*)aux; p.ld_aux = ld_aux;
    p.resid = resid; p.ld_resid = ld_resid; p.period = period;
    p.drop_thr = drop_rate > 0.0f ? chb_drop_threshold(drop_rate) : 0u;
    p.drop_scale = 1.0f / (1.0f - drop_rate);
    p.drop_key = drop_key;
    p.tiles_m = chb_div_up(M, BM); p.tiles_n = chb_div_up(N, BN);
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case CHB_EPI_NONE: launch_nt<CHB_EPI_NONE>(p, out_dtype, s); break;
        case CHB_EPI_GELU: launch_nt<CHB_EPI_GELU>(p, out_dtype, s); break;
        case CHB_EPI_DGELU: launch_nt<CHB_EPI_DGELU>(p, out_dtype, s); break;
        case CHB_EPI_RESID: launch_nt<CHB_EPI_RESID>(p, out_dtype, s); break;
        case CHB_EPI_PATCH: launch_nt<CHB_EPI_PATCH>(p, out_dtype, s); break;
        default: return CHB_EINVAL;
    }
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_gemm_tn(const void* X, int64_t ldx, const void* dY, int64_t ldy, float* dW, int64_t ldw, int M, int Kd, int Nd,
                void* stream) {
    if (!X || !dY || !dW || M < 0 || Kd <= 0 || Nd <= 0) return CHB_EINVAL;
    if (M == 0) return CHB_OK;
    if (M % 64 != 0 || (Kd & 7) || (Nd & 7) || (ldx & 7) || (ldy & 7)) return CHB_EUNSUPPORTED;
    if (((uintptr_t)X & 15) || ((uintptr_t)dY & 15)) return CHB_EINVAL;
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
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * p.splits), dim3(256), 0, (hipStream_t)stream, p);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

}  // extern "C"
