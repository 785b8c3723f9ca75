// RandAugment / AutoAugment primitive ops on uint8 NHWC batches, gfx950.
//
// Every kernel here is HBM-bound byte work: one pass reads each input byte once and
// writes each output byte once (two-pass ops read the input twice).  Threads own
// 4 pixels = 12 contiguous bytes (global_load_dwordx3 / global_store_dwordx3), so a
// wave touches 768 contiguous bytes per instruction and RGB triples never straddle
// threads.  Compiled with -ffp-contract=off: the float-mediated ops must round after
// every multiply and add exactly like the un-fused TF CPU kernels the oracle restates.
//
// Reference semantics: chambers/augmentations/image_augmentations.py (lines cited per
// kernel); upstream TF / tensorflow-addons behaviour as restated in oracle/augment_ref.py.
#include "common.hpp"
#include <mutex>
#include <type_traits>
#include <vector>
#include "../../include/chambers_hip.h"
#include <string.h>

namespace {

struct __attribute__((packed, aligned(4))) px4_t { uint32_t w[3]; };

__device__ __forceinline__ void unpack12(const px4_t& p, uint8_t (&b)[12]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        b[4 * i + 0] = p.w[i] & 0xff;
        b[4 * i + 1] = (p.w[i] >> 8) & 0xff;
        b[4 * i + 2] = (p.w[i] >> 16) & 0xff;
        b[4 * i + 3] = (p.w[i] >> 24) & 0xff;
    }
}
__device__ __forceinline__ px4_t pack12(const uint8_t (&b)[12]) {
    px4_t p;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        p.w[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) |
                 ((uint32_t)b[4 * i + 3] << 24);
    return p;
}

// tf.cast(float -> uint8) of an in-range value: truncation
__device__ __forceinline__ uint8_t trunc_u8(float v) { return (uint8_t)(int)v; }

// blend(), image_augmentations.py:10-49, for factor not in {0, 1}
template <bool CLIP>
__device__ __forceinline__ uint8_t blend1(uint8_t deg, uint8_t x, float factor) {
    const float i1 = (float)deg, i2 = (float)x;
    const float diff = i2 - i1;
    const float scaled = factor * diff;
    float t = i1 + scaled;
    if (CLIP) t = fminf(fmaxf(t, 0.0f), 255.0f);
    return trunc_u8(t);
}

// tf.image.rgb_to_grayscale on uint8 (see oracle rgb_to_grayscale)
__device__ __forceinline__ uint8_t gray_u8(uint8_t r, uint8_t g, uint8_t b) {
    const float s = 1.0f / 255.0f;
    float v = ((float)r * s) * 0.2989f;
    v = v + ((float)g * s) * 0.5870f;
    v = v + ((float)b * s) * 0.1140f;
    v = v * 255.5f;
    return trunc_u8(v);
}

struct PwParams {
    int op;
    float factor;
    int i0, i1;
};

template <int OP>
__device__ __forceinline__ void pointwise12(uint8_t (&b)[12], const PwParams& pp) {
    if (OP == CHB_PW_INVERT) {  // :112-113
#pragma unroll
        for (int i = 0; i < 12; ++i) b[i] = 255 - b[i];
    } else if (OP == CHB_PW_POSTERIZE) {  // :171-174, i0 = shift
#pragma unroll
        for (int i = 0; i < 12; ++i) b[i] = (uint8_t)((b[i] >> pp.i0) << pp.i0);
    } else if (OP == CHB_PW_SOLARIZE) {  // :192-193, i0 = threshold
#pragma unroll
        for (int i = 0; i < 12; ++i) b[i] = ((int)b[i] < pp.i0) ? b[i] : (uint8_t)(255 - b[i]);
    } else if (OP == CHB_PW_SOLARIZE_ADD) {  // :212-215, i0 = threshold, i1 = addition
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            int v = (int)b[i] + pp.i1;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            b[i] = ((int)b[i] < pp.i0) ? (uint8_t)v : b[i];
        }
    } else if (OP == CHB_PW_BRIGHTNESS) {  // :283-285
        const bool clip = !(pp.factor > 0.0f && pp.factor < 1.0f);
#pragma unroll
        for (int i = 0; i < 12; ++i) b[i] = clip ? blend1<true>(0, b[i], pp.factor) : blend1<false>(0, b[i], pp.factor);
    } else if (OP == CHB_PW_CONTRAST) {  // :253-265, i0 = degenerate constant
        const bool clip = !(pp.factor > 0.0f && pp.factor < 1.0f);
        const uint8_t d = (uint8_t)pp.i0;
#pragma unroll
        for (int i = 0; i < 12; ++i) b[i] = clip ? blend1<true>(d, b[i], pp.factor) : blend1<false>(d, b[i], pp.factor);
    } else if (OP == CHB_PW_COLOR) {  // :233-235 (3 channels)
        const bool clip = !(pp.factor > 0.0f && pp.factor < 1.0f);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint8_t d = gray_u8(b[3 * p], b[3 * p + 1], b[3 * p + 2]);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                b[3 * p + c] = clip ? blend1<true>(d, b[3 * p + c], pp.factor) : blend1<false>(d, b[3 * p + c], pp.factor);
        }
    }
}

template <int OP>
__global__ void __launch_bounds__(256) pointwise_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                        int64_t n_groups, int64_t n_bytes, PwParams pp) {
    const px4_t* in4 = reinterpret_cast<const px4_t*>(in);
    px4_t* out4 = reinterpret_cast<px4_t*>(out);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += stride) {
        px4_t p = in4[g];
        uint8_t b[12];
        unpack12(p, b);
        pointwise12<OP>(b, pp);
        out4[g] = pack12(b);
    }
    // tail (< 12 bytes, whole pixels): one thread
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t base = n_groups * 12;
        const int tail = (int)(n_bytes - base);
        if (tail > 0) {
            uint8_t b[12];
            for (int i = 0; i < 12; ++i) b[i] = i < tail ? in[base + i] : 0;
            pointwise12<OP>(b, pp);
            for (int i = 0; i < tail; ++i) out[base + i] = b[i];
        }
    }
}

inline int row_grid(int64_t rows) {   // one wave per image row, 4 rows per block
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    return (int)(blocks < 1 ? 1 : blocks);
}

inline int stream_grid(int64_t n_groups) {
    int64_t blocks = (n_groups + 255) / 256;
    if (blocks > 8192) blocks = 8192;  // 256 CUs x 8 blocks x 4: grid-stride the rest
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ---- nearest-neighbour projective warp (tfa.image.transform), :333-341 etc. -----
// One wave per output row (no per-thread integer division), a lane owns 4 output pixels; each source pixel is
// fetched with ONE unaligned dword load (3 payload bytes) instead of three byte loads.
struct __attribute__((packed)) u32_unaligned { uint32_t v; };

__global__ void __launch_bounds__(256) affine_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int H,
                                                     int W, int C, const float* __restrict__ tdev, int per_image,
                                                     float t0, float t1, float t2, float t3, float t4, float t5,
                                                     float t6, float t7, int fill) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wq = (W + 3) >> 2;  // x-quads per row
    const int rows = B * H;
    const int64_t total_bytes = (int64_t)B * H * W * C;
    const bool fast = (C == 3) && ((W & 3) == 0);
    // one wave per 4 consecutive output rows: 16 independent gathers in flight per lane
    for (int row0 = (blockIdx.x * 4 + wave) * 4; row0 < rows; row0 += gridDim.x * 16) {
        for (int xq = lane; xq < wq; xq += 64) {
            const int x0 = xq * 4;
            uint32_t w[4][4];
            bool valid[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = row0 + k;
                valid[k] = row < rows;
                const int rr = valid[k] ? row : rows - 1;
                const int n = rr / H, oy = rr - n * H;
                float a0 = t0, a1 = t1, a2 = t2, b0 = t3, b1 = t4, b2 = t5, c0 = t6, c1 = t7;
                if (tdev) {
                    const float* t = tdev + (per_image ? (int64_t)n * 8 : 0);
                    a0 = t[0]; a1 = t[1]; a2 = t[2]; b0 = t[3]; b1 = t[4]; b2 = t[5]; c0 = t[6]; c1 = t[7];
                }
                const int64_t img_off = (int64_t)n * H * W * C;
                const uint8_t* img = in + img_off;
                const float fy = (float)oy;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float fx = (float)(x0 + i);
                    // affine rows (c0 == c1 == 0, every op of the augmentation schemes): proj is exactly 1 and x / 1 == x, so the
                    // two IEEE divisions are skipped without changing a bit; true projective rows take the division path
                    const bool affine_only = (c0 == 0.0f) && (c1 == 0.0f);
                    const float proj = affine_only ? 1.0f : (c0 * fx + c1 * fy) + 1.0f;
                    float ix = (a0 * fx + a1 * fy) + a2;
                    float iy = (b0 * fx + b1 * fy) + b2;
                    if (!affine_only) {
                        ix = ix / proj;
                        iy = iy / proj;
                    }
                    const float rx = roundf(ix), ry = roundf(iy);  // half away from zero
                    const bool ok = (proj != 0.0f) && (rx >= 0.0f) && (rx < (float)W) && (ry >= 0.0f) && (ry < (float)H);
                    const int64_t src = ok ? ((int64_t)(int)ry * W + (int)rx) * C : 0;
                    uint32_t v = ((uint32_t)fill) * 0x01010101u;
                    if (ok) {
                        if (C == 3) {
                            if (img_off + src + 4 <= total_bytes) v = reinterpret_cast<const u32_unaligned*>(img + src)->v;
                            else v = (uint32_t)img[src] | ((uint32_t)img[src + 1] << 8) | ((uint32_t)img[src + 2] << 16);
                        } else {
                            v = 0;
                            for (int c = 0; c < C; ++c) v |= (uint32_t)img[src + c] << (8 * c);
                        }
                    }
                    w[k][i] = v;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!valid[k]) break;
                uint8_t* orow = out + (int64_t)(row0 + k) * W * C;
                if (fast) {
                    uint8_t b[12];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        b[3 * i + 0] = w[k][i] & 0xff; b[3 * i + 1] = (w[k][i] >> 8) & 0xff; b[3 * i + 2] = (w[k][i] >> 16) & 0xff;
                    }
                    *reinterpret_cast<px4_t*>(orow + (int64_t)x0 * 3) = pack12(b);
                } else {
                    for (int i = 0; i < 4; ++i)
                        if (x0 + i < W)
                            for (int c = 0; c < C; ++c) orow[(int64_t)(x0 + i) * C + c] = (uint8_t)(w[k][i] >> (8 * c));
                }
            }
        }
    }
}

// Fast path for what the augmentation schemes launch (one affine transform for the whole batch, projective row = 0, RGB,
// W % 4 == 0): no division, no per-pixel branch.  A wave owns 4 consecutive output rows, a lane 4 pixels of each; the per-row
// terms a1*y, b1*y are formed once (same fp32 operation order as the generic kernel: (a0*x + a1*y) + a2), every source pixel
// is one unaligned dword load from a clamped address, and `fill` is selected afterwards.
__global__ void __launch_bounds__(256) affine_rgb_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int H,
                                                         int W, float a0, float a1, float a2, float b0, float b1, float b2, int fill) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wq = W >> 2;
    const int rows = B * H;
    const int64_t img_bytes = (int64_t)H * W * 3;
    const uint32_t fillw = ((uint32_t)fill) * 0x01010101u;
    const float fW = (float)W, fH = (float)H;
    for (int row0 = (blockIdx.x * 4 + wave) * 4; row0 < rows; row0 += gridDim.x * 16) {
        for (int xq = lane; xq < wq; xq += 64) {
            const int x0 = xq * 4;
            float fx[4], ax[4], bx[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fx[i] = (float)(x0 + i);
                ax[i] = a0 * fx[i];
                bx[i] = b0 * fx[i];
            }
            uint32_t w[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rr = min(row0 + k, rows - 1);   // wave-uniform
                const int n = rr / H, oy = rr - n * H;
                const uint8_t* img = in + (int64_t)n * img_bytes;
                const bool last_img = n == B - 1;
                const float fy = (float)oy;
                const float ay = a1 * fy, by = b1 * fy;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float ix = (ax[i] + ay) + a2;
                    const float iy = (bx[i] + by) + b2;
                    const float rx = roundf(ix), ry = roundf(iy);  // half away from zero
                    const bool ok = (rx >= 0.0f) && (rx < fW) && (ry >= 0.0f) && (ry < fH);
                    const int off = ok ? ((int)ry * W + (int)rx) * 3 : 0;
                    // the dword at the very last pixel of the batch would end one byte past the buffer: step back and shift
                    const int over = (last_img && (int64_t)off + 4 > img_bytes) ? 1 : 0;
                    const uint32_t v = reinterpret_cast<const u32_unaligned*>(img + off - over)->v >> (8 * over);
                    w[k][i] = ok ? v : fillw;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (row0 + k >= rows) break;
                uint8_t b[12];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    b[3 * i + 0] = w[k][i] & 0xff; b[3 * i + 1] = (w[k][i] >> 8) & 0xff; b[3 * i + 2] = (w[k][i] >> 16) & 0xff;
                }
                *reinterpret_cast<px4_t*>(out + (int64_t)(row0 + k) * W * 3 + (int64_t)x0 * 3) = pack12(b);
            }
        }
    }
}

// The same transform over TW x (256/TW) output tiles for transforms that mix rows (rotation): one wave per tile, lane =
// (row group r, column x) owns the pixels (x, r + RG*j), j = 0..3, RG = 64/TW.  One gather instruction then covers a TW x RG block
// whose source footprint is a rotated TW x RG block (~11 rows of one or two lines for 27 degrees at TW = 16) instead of the 64
// separate lines a row-long instruction touches under rotation; the tile is transposed through a wave-private 1 KiB LDS block so
// that a lane stores 4 consecutive pixels (12 bytes).
template <int TW>
__global__ void __launch_bounds__(256) affine_rgb_tile_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int H,
                                                              int W, float a0, float a1, float a2, float b0, float b1, float b2, int fill) {
    constexpr int TH = 256 / TW, RG = 64 / TW, QW = TW / 4;
    __shared__ uint32_t tile[4][256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;
    const int tiles = B * ty * tx;
    const int64_t img_bytes = (int64_t)H * W * 3;
    const uint32_t fillw = ((uint32_t)fill) * 0x01010101u;
    const float fW = (float)W, fH = (float)H;
    const int rg = lane / TW, xl = lane % TW;
    const int orow = lane / QW, oq = lane % QW;     // store role: row of the tile, 4-pixel quad
    uint32_t* lds = tile[wave];
    for (int t = blockIdx.x * 4 + wave; t < tiles; t += gridDim.x * 4) {
        const int n = t / (ty * tx), rem = t - n * (ty * tx);
        const int y0 = (rem / tx) * TH, x0 = (rem - (rem / tx) * tx) * TW;
        const uint8_t* img = in + (int64_t)n * img_bytes;
        const bool last_img = n == B - 1;
        const float fx = (float)(x0 + xl);
        const float ax = a0 * fx, bx = b0 * fx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float fy = (float)(y0 + rg + RG * j);
            const float ix = (ax + a1 * fy) + a2;            // (a0*x + a1*y) + a2, as the row kernel
            const float iy = (bx + b1 * fy) + b2;
            const float rx = roundf(ix), ry = roundf(iy);
            const bool ok = (rx >= 0.0f) && (rx < fW) && (ry >= 0.0f) && (ry < fH);
            const int off = ok ? ((int)ry * W + (int)rx) * 3 : 0;
            const int over = (last_img && (int64_t)off + 4 > img_bytes) ? 1 : 0;
            const uint32_t v = reinterpret_cast<const u32_unaligned*>(img + off - over)->v >> (8 * over);
            lds[(rg + RG * j) * TW + xl] = ok ? v : fillw;
        }
        // wave-private block: every lane's writes land before the wave's reads (LDS operations of a wave complete in order)
        const uint4 q = *reinterpret_cast<const uint4*>(lds + orow * TW + oq * 4);
        const int oy = y0 + orow, ox = x0 + oq * 4;
        if (oy < H && ox < W) {
            px4_t o;
            o.w[0] = (q.x & 0xffffffu) | (q.y << 24);
            o.w[1] = ((q.y >> 8) & 0xffffu) | (q.z << 16);
            o.w[2] = ((q.z >> 16) & 0xffu) | (q.w << 8);
            *reinterpret_cast<px4_t*>(out + ((int64_t)n * H + oy) * W * 3 + (int64_t)ox * 3) = o;
        }
    }
}

// ---- cutout (tfa.image.random_cutout with explicit centres), :495-499 -------------
// One wave per 4 consecutive image rows (4 independent 12-byte loads in flight per lane, no per-thread division).
__global__ void __launch_bounds__(256) cutout_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int H,
                                                     int W, int C, const int32_t* __restrict__ centers, int half,
                                                     int value) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wq = (W + 3) >> 2;
    const int rows = B * H;
    for (int row0 = (blockIdx.x * 4 + wave) * 4; row0 < rows; row0 += gridDim.x * 16) {
        if (C == 3 && (W & 3) == 0) {
            for (int xq = lane; xq < wq; xq += 64) {
                const int x0 = xq * 4;
                px4_t p[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (row0 + k < rows) p[k] = *reinterpret_cast<const px4_t*>(in + (int64_t)(row0 + k) * W * 3 + (int64_t)x0 * 3);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = row0 + k;
                    if (row >= rows) break;
                    const int n = row / H, y = row - n * H;
                    const int cy = centers[2 * n], cx = centers[2 * n + 1];
                    const int xa = max(0, cx - half), xb = min(W, cx + half);
                    const bool rowin = (y >= max(0, cy - half)) && (y < min(H, cy + half));
                    if (rowin && x0 + 3 >= xa && x0 < xb) {
                        uint8_t b[12];
                        unpack12(p[k], b);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool inside = (x0 + i >= xa) && (x0 + i < xb);
#pragma unroll
                            for (int c = 0; c < 3; ++c) b[3 * i + c] = inside ? (uint8_t)value : b[3 * i + c];
                        }
                        p[k] = pack12(b);
                    }
                    *reinterpret_cast<px4_t*>(out + (int64_t)row * W * 3 + (int64_t)x0 * 3) = p[k];
                }
            }
        } else {
            for (int k = 0; k < 4; ++k) {
                const int row = row0 + k;
                if (row >= rows) break;
                const int n = row / H, y = row - n * H;
                const int cy = centers[2 * n], cx = centers[2 * n + 1];
                const int xa = max(0, cx - half), xb = min(W, cx + half);
                const bool rowin = (y >= max(0, cy - half)) && (y < min(H, cy + half));
                const int64_t rowoff = (int64_t)row * W * C;
                for (int x = lane; x < W; x += 64) {
                    const bool inside = rowin && (x >= xa) && (x < xb);
                    for (int c = 0; c < C; ++c) {
                        const int64_t o = rowoff + (int64_t)x * C + c;
                        out[o] = inside ? (uint8_t)value : in[o];
                    }
                }
            }
        }
    }
}

// ---- per-image, per-channel statistics (AutoContrast min/max, Equalize histogram) ----
__global__ void stats_init_kernel(int32_t* ws, int n, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ws[i] = (mode == 0) ? ((i & 1) ? 0 : 255) : 0;  // mode 0: [min,max] pairs; mode 1: histogram
}

// grid = (slices, B).  C == 3 fast path reads 12-byte groups; generic path reads bytes.
__global__ void __launch_bounds__(256) minmax_kernel(const uint8_t* __restrict__ in, int32_t* __restrict__ ws, int HW, int C) {
    const int n = blockIdx.y;
    const uint8_t* img = in + (int64_t)n * HW * C;
    int lo[4] = {255, 255, 255, 255}, hi[4] = {0, 0, 0, 0};
    const int64_t nbytes = (int64_t)HW * C;
    if (C == 3) {
        const int64_t ng = nbytes / 12;
        const px4_t* p4 = reinterpret_cast<const px4_t*>(img);
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += (int64_t)gridDim.x * blockDim.x) {
            uint8_t b[12];
            unpack12(p4[g], b);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                lo[i % 3] = min(lo[i % 3], (int)b[i]);
                hi[i % 3] = max(hi[i % 3], (int)b[i]);
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (int64_t i = ng * 12; i < nbytes; ++i) {
                lo[i % 3] = min(lo[i % 3], (int)img[i]);
                hi[i % 3] = max(hi[i % 3], (int)img[i]);
            }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += (int64_t)gridDim.x * blockDim.x) {
            const int c = (int)(i % C);
            lo[c] = min(lo[c], (int)img[i]);
            hi[c] = max(hi[c], (int)img[i]);
        }
    }
    for (int c = 0; c < C; ++c) {
        int l = lo[c], h = hi[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            l = min(l, __shfl_xor(l, o, 64));
            h = max(h, __shfl_xor(h, o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&ws[(n * C + c) * 2 + 0], l);
            atomicMax(&ws[(n * C + c) * 2 + 1], h);
        }
    }
}

// AutoContrast apply, :72-86
__global__ void __launch_bounds__(256) autocontrast_apply_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                                 const int32_t* __restrict__ ws, int HW, int C) {
    const int n = blockIdx.y;
    float scale[4], offset[4];
    for (int c = 0; c < C; ++c) {
        const float lo = (float)ws[(n * C + c) * 2], hi = (float)ws[(n * C + c) * 2 + 1];
        const float rng = hi - lo;
        float s = (rng != 0.0f) ? 255.0f / rng : 0.0f;  // divide_no_nan
        float o = (-lo) * s;
        const float mask = hi > lo ? 1.0f : 0.0f;
        s = s * mask + (1.0f - mask);
        o = o * mask;
        scale[c] = s;
        offset[c] = o;
    }
    const int64_t base = (int64_t)n * HW * C;
    const int64_t nbytes = (int64_t)HW * C;
    if (C == 3) {
        const int64_t ng = nbytes / 12;
        const px4_t* p4 = reinterpret_cast<const px4_t*>(in + base);
        px4_t* o4 = reinterpret_cast<px4_t*>(out + base);
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += (int64_t)gridDim.x * blockDim.x) {
            uint8_t b[12];
            unpack12(p4[g], b);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                float v = (float)b[i] * scale[i % 3];
                v = v + offset[i % 3];
                v = fminf(fmaxf(v, 0.0f), 255.0f);
                b[i] = trunc_u8(v);
            }
            o4[g] = pack12(b);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (int64_t i = ng * 12; i < nbytes; ++i) {
                float v = (float)in[base + i] * scale[i % 3];
                v = v + offset[i % 3];
                v = fminf(fmaxf(v, 0.0f), 255.0f);
                out[base + i] = trunc_u8(v);
            }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += (int64_t)gridDim.x * blockDim.x) {
            const int c = (int)(i % C);
            float v = (float)in[base + i] * scale[c];
            v = v + offset[c];
            v = fminf(fmaxf(v, 0.0f), 255.0f);
            out[base + i] = trunc_u8(v);
        }
    }
}

// Equalize pass 1: per-block LDS histogram (C <= 4), merged with global atomics.
__global__ void __launch_bounds__(256) hist_kernel(const uint8_t* __restrict__ in, int32_t* __restrict__ ws, int HW, int C) {
    __shared__ int32_t h[4 * 256];
    for (int i = threadIdx.x; i < C * 256; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const int n = blockIdx.y;
    const uint8_t* img = in + (int64_t)n * HW * C;
    const int64_t nbytes = (int64_t)HW * C;
    if (C == 3) {
        const int64_t ng = nbytes / 12;
        const px4_t* p4 = reinterpret_cast<const px4_t*>(img);
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += (int64_t)gridDim.x * blockDim.x) {
            uint8_t b[12];
            unpack12(p4[g], b);
#pragma unroll
            for (int i = 0; i < 12; ++i) atomicAdd(&h[(i % 3) * 256 + b[i]], 1);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (int64_t i = ng * 12; i < nbytes; ++i) atomicAdd(&h[(i % 3) * 256 + img[i]], 1);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += (int64_t)gridDim.x * blockDim.x)
            atomicAdd(&h[(int)(i % C) * 256 + img[i]], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 256; i += blockDim.x)
        if (h[i]) atomicAdd(&ws[(int64_t)n * C * 256 + i], h[i]);
}

// Equalize pass 2: histogram -> LUT in place (tfa.image.equalize _scale_channel).
// grid = B*C blocks of 256 threads; bin i owned by thread i.
__global__ void __launch_bounds__(256) equalize_lut_kernel(int32_t* __restrict__ ws) {
    __shared__ int32_t s[256];
    __shared__ int32_t last_nz;
    int32_t* h = ws + (int64_t)blockIdx.x * 256;
    const int t = threadIdx.x;
    const int32_t mine = h[t];
    s[t] = mine;
    if (t == 0) last_nz = 0;
    __syncthreads();
    if (mine != 0) atomicMax(&last_nz, t);
    // inclusive scan (Hillis-Steele) over 256 bins
    for (int o = 1; o < 256; o <<= 1) {
        const int32_t v = (t >= o) ? s[t - o] : 0;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    const int32_t total = s[255];
    const int32_t excl = s[t] - mine;
    const int32_t step = (total - h[last_nz]) / 255;
    __syncthreads();
    int32_t lut = t;
    if (step != 0) {
        lut = (excl + step / 2) / step;
        lut = lut < 0 ? 0 : (lut > 255 ? 255 : lut);
    }
    h[t] = lut;
}

// Equalize pass 3: out = lut[n][c][in]
__global__ void __launch_bounds__(256) lut_apply_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                        const int32_t* __restrict__ ws, int HW, int C) {
    __shared__ uint8_t lut[4 * 256];
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < C * 256; i += blockDim.x) lut[i] = (uint8_t)ws[(int64_t)n * C * 256 + i];
    __syncthreads();
    const int64_t base = (int64_t)n * HW * C;
    const int64_t nbytes = (int64_t)HW * C;
    if (C == 3) {
        const int64_t ng = nbytes / 12;
        const px4_t* p4 = reinterpret_cast<const px4_t*>(in + base);
        px4_t* o4 = reinterpret_cast<px4_t*>(out + base);
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += (int64_t)gridDim.x * blockDim.x) {
            uint8_t b[12];
            unpack12(p4[g], b);
#pragma unroll
            for (int i = 0; i < 12; ++i) b[i] = lut[(i % 3) * 256 + b[i]];
            o4[g] = pack12(b);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (int64_t i = ng * 12; i < nbytes; ++i) out[base + i] = lut[(i % 3) * 256 + in[base + i]];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += (int64_t)gridDim.x * blockDim.x)
            out[base + i] = lut[(int)(i % C) * 256 + in[base + i]];
    }
}

// ---- Sharpness (tfa.image.sharpness), :303-304 ---------------------------------------
// One wave per row; a lane owns 4 output pixels and pulls the 3 x 6-pixel window it needs as 3 x 5 unaligned dword
// loads (18 payload bytes per row) instead of 108 byte loads.  The 1-pixel border keeps the original.
template <int MODE>  // 0: factor==0 (degenerate only), 1: 0<f<1 (no clip), 2: clip
__device__ __forceinline__ uint8_t sharp_finish(uint8_t deg, uint8_t orig, float factor) {
    if (MODE == 0) return deg;
    if (MODE == 1) return blend1<false>(deg, orig, factor);
    return blend1<true>(deg, orig, factor);
}

template <int MODE>
__global__ void __launch_bounds__(256) sharpness_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int H,
                                                        int W, int C, float factor) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wq = (W + 3) >> 2;
    const int rows = B * H;
    const int64_t total_bytes = (int64_t)B * H * W * C;
    const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const int n = row / H, y = row - n * H;
        const uint8_t* img = in + (int64_t)n * H * W * C;
        const int64_t rowoff = (int64_t)row * W * C;
        const bool yin = (y >= 1) && (y < H - 1);
        for (int xq = lane; xq < wq; xq += 64) {
            const int x0 = xq * 4;
            if (C == 3 && (W & 3) == 0) {
                // window bytes [s, s + 20) of rows y-1, y, y+1 with s = (x0 - 1) * 3 (relative to the row start)
                uint8_t w[3][20];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int yy = yin ? y + r - 1 : y;
                    const int64_t base = ((int64_t)n * H + yy) * (int64_t)W * 3 + (int64_t)(x0 - 1) * 3;   // offset in the whole batch
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        const int64_t o = base + 4 * k;
                        uint32_t v = 0;
                        if (o >= 0 && o + 4 <= total_bytes) v = reinterpret_cast<const u32_unaligned*>(in + o)->v;
                        else { for (int bb = 0; bb < 4; ++bb) if (o + bb >= 0 && o + bb < total_bytes) v |= (uint32_t)in[o + bb] << (8 * bb); }
                        w[r][4 * k + 0] = v & 0xff; w[r][4 * k + 1] = (v >> 8) & 0xff; w[r][4 * k + 2] = (v >> 16) & 0xff; w[r][4 * k + 3] = v >> 24;
                    }
                }
                uint8_t b[12];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int x = x0 + i;
                    const bool interior = yin && (x >= 1) && (x < W - 1);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const uint8_t orig = w[1][(i + 1) * 3 + c];
                        uint8_t deg = orig;
                        if (interior) {
                            float acc = 0.0f;
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                                for (int kx = 0; kx < 3; ++kx) {
                                    const float v = (float)w[ky][(i + kx) * 3 + c];
                                    acc = acc + v * ((ky == 1 && kx == 1) ? k5 : k1);
                                }
                            deg = trunc_u8(acc);
                        }
                        b[3 * i + c] = sharp_finish<MODE>(deg, orig, factor);
                    }
                }
                *reinterpret_cast<px4_t*>(out + rowoff + (int64_t)x0 * 3) = pack12(b);
            } else {
                for (int i = 0; i < 4; ++i) {
                    const int x = x0 + i;
                    if (x >= W) break;
                    const bool interior = yin && (x >= 1) && (x < W - 1);
                    for (int c = 0; c < C; ++c) {
                        const uint8_t orig = img[((int64_t)y * W + x) * C + c];
                        uint8_t deg = orig;
                        if (interior) {
                            float acc = 0.0f;
                            for (int ky = -1; ky <= 1; ++ky)
                                for (int kx = -1; kx <= 1; ++kx) {
                                    const float v = (float)img[((int64_t)(y + ky) * W + (x + kx)) * C + c];
                                    acc = acc + v * ((ky == 0 && kx == 0) ? k5 : k1);
                                }
                            deg = trunc_u8(acc);
                        }
                        out[rowoff + (int64_t)x * C + c] = sharp_finish<MODE>(deg, orig, factor);
                    }
                }
            }
        }
    }
}

// Fast path (C == 3, W % 4 == 0): one wave per PAIR of output rows.  A lane owns the same 4-pixel quad in both rows; the four
// input rows it needs arrive as aligned 12-byte loads (one per row), the left / right halo pixel comes from the neighbouring
// lane's registers (a direct dword load only where the 64-lane window ends inside the row).  Every input byte is converted and
// multiplied by 1/13 once and reused by up to six outputs; the per-output sum keeps the oracle's row-major order, so the result
// is bit-identical to the generic kernel.
template <int MODE>
__global__ void __launch_bounds__(256) sharpness_rows_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int H,
                                                             int W, float factor) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wq = W >> 2;
    const int hp = (H + 1) >> 1;                 // row pairs per image
    const int pairs = B * hp;
    const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
    const int64_t row_bytes = (int64_t)W * 3;
    for (int pr = blockIdx.x * 4 + wave; pr < pairs; pr += gridDim.x * 4) {
        const int n = pr / hp, y0 = (pr - n * hp) * 2;          // output rows y0, y0 + 1 of image n
        const uint8_t* img = in + (int64_t)n * H * row_bytes;
        uint8_t* oimg = out + (int64_t)n * H * row_bytes;
        for (int xb = 0; xb < wq; xb += 64) {
            const int xq = xb + lane;
            const bool live = xq < wq;
            const int x0 = xq * 4;
            // input rows y0-1 .. y0+2 (rows outside the image are never used: border rows keep the original)
            float p1[4][18];     // (float)byte * k1 for [left px | own 4 px | right px]
            uint32_t own[2][3];  // the original bytes of the two output rows
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int yy = y0 - 1 + r;
                const bool rin = (yy >= 0) && (yy < H);
                px4_t v;
                v.w[0] = v.w[1] = v.w[2] = 0;
                const uint8_t* rp = img + (int64_t)(rin ? yy : 0) * row_bytes;
                if (rin && live) v = *reinterpret_cast<const px4_t*>(rp + (int64_t)x0 * 3);
                uint32_t lft = __shfl_up(v.w[2], 1, 64), rgt = __shfl_down(v.w[0], 1, 64);
                if (rin && live && lane == 0 && xq > 0) lft = *reinterpret_cast<const uint32_t*>(rp + (int64_t)x0 * 3 - 4);
                if (rin && live && lane == 63 && xq + 1 < wq) rgt = *reinterpret_cast<const uint32_t*>(rp + (int64_t)x0 * 3 + 12);
                if (r == 1 || r == 2) { own[r - 1][0] = v.w[0]; own[r - 1][1] = v.w[1]; own[r - 1][2] = v.w[2]; }
                p1[r][0] = (float)((lft >> 8) & 0xff) * k1;
                p1[r][1] = (float)((lft >> 16) & 0xff) * k1;
                p1[r][2] = (float)(lft >> 24) * k1;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    p1[r][3 + 4 * k + 0] = (float)(v.w[k] & 0xff) * k1;
                    p1[r][3 + 4 * k + 1] = (float)((v.w[k] >> 8) & 0xff) * k1;
                    p1[r][3 + 4 * k + 2] = (float)((v.w[k] >> 16) & 0xff) * k1;
                    p1[r][3 + 4 * k + 3] = (float)(v.w[k] >> 24) * k1;
                }
                p1[r][15] = (float)(rgt & 0xff) * k1;
                p1[r][16] = (float)((rgt >> 8) & 0xff) * k1;
                p1[r][17] = (float)((rgt >> 16) & 0xff) * k1;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = y0 + j;
                if (y >= H || !live) continue;
                const bool yin = (y >= 1) && (y < H - 1);
                uint8_t ob[12];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    ob[4 * k + 0] = own[j][k] & 0xff; ob[4 * k + 1] = (own[j][k] >> 8) & 0xff;
                    ob[4 * k + 2] = (own[j][k] >> 16) & 0xff; ob[4 * k + 3] = own[j][k] >> 24;
                }
                uint8_t b[12];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int x = x0 + i;
                    const bool interior = yin && (x >= 1) && (x < W - 1);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const uint8_t orig = ob[3 * i + c];
                        // window column (i + kx) of p1 is pixel x - 1 + kx; rows j, j+1, j+2 are y-1, y, y+1
                        float acc = p1[j][3 * i + c];
                        acc = acc + p1[j][3 * (i + 1) + c];
                        acc = acc + p1[j][3 * (i + 2) + c];
                        acc = acc + p1[j + 1][3 * i + c];
                        acc = acc + (float)orig * k5;
                        acc = acc + p1[j + 1][3 * (i + 2) + c];
                        acc = acc + p1[j + 2][3 * i + c];
                        acc = acc + p1[j + 2][3 * (i + 1) + c];
                        acc = acc + p1[j + 2][3 * (i + 2) + c];
                        const uint8_t deg = interior ? trunc_u8(acc) : orig;
                        b[3 * i + c] = sharp_finish<MODE>(deg, orig, factor);
                    }
                }
                *reinterpret_cast<px4_t*>(oimg + (int64_t)y * row_bytes + (int64_t)x0 * 3) = pack12(b);
            }
        }
    }
}

// ---- ImageNetNormalization, :629-682 ---------------------------------------------------
struct NormConst { float mean[3]; float stdv[3]; };

template <int MODE, typename TIN>  // 0 caffe, 1 tf, 2 torch
__device__ __forceinline__ float norm1(TIN xin, int c, const NormConst& nc) {
    const float x = (float)xin;
    if (MODE == 1) {
        float v = x / 127.5f;
        return v - 1.0f;
    } else if (MODE == 2) {
        float v = x / 255.0f;
        v = v - nc.mean[c];
        return v / nc.stdv[c];
    } else {
        return x - nc.mean[c];
    }
}

template <int MODE, typename TIN>
__global__ void __launch_bounds__(256) normalize_kernel(const TIN* __restrict__ in, float* __restrict__ out, int64_t n_pixels, NormConst nc) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pixels; p += (int64_t)gridDim.x * blockDim.x) {
        const TIN* s = in + p * 3;
        float* d = out + p * 3;
        if (MODE == 0) {  // RGB -> BGR then subtract mean (:647-650)
            const TIN r = s[0], g = s[1], b = s[2];
            d[0] = norm1<0>(b, 0, nc);
            d[1] = norm1<0>(g, 1, nc);
            d[2] = norm1<0>(r, 2, nc);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = norm1<MODE>(s[c], c, nc);
        }
    }
}

// modes "tf" and "torch" are elementwise given the channel of each byte: a lane converts one aligned dword (4 consecutive
// bytes) into one float4, so every wave-level load is 256 contiguous bytes and every store 1 KiB contiguous.
template <int MODE>
__global__ void __launch_bounds__(256) normalize_u8x4_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t n_quads,
                                                             int64_t n_bytes, int C, NormConst nc) {
    const uint32_t* in4 = reinterpret_cast<const uint32_t*>(in);
    float4* out4 = reinterpret_cast<float4*>(out);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q0 < n_quads; q0 += 4 * stride) {
        uint32_t w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t q = q0 + u * stride;
            w[u] = q < n_quads ? in4[q] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t q = q0 + u * stride;
            if (q >= n_quads) break;
            const int c0 = MODE == 1 ? 0 : (int)((q * 4) % C);
            float f[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int c = c0 + k;
                if (MODE != 1) c = c >= C ? c - C : c, c = c >= C ? c - C : c;   // C == 3 here: at most two wraps
                f[k] = norm1<MODE>((uint8_t)(w[u] >> (8 * k)), c, nc);
            }
            out4[q] = make_float4(f[0], f[1], f[2], f[3]);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t e = n_quads * 4; e < n_bytes; ++e) out[e] = norm1<MODE>(in[e], (int)(e % C), nc);
}

// Normalise (mode tf/torch/caffe) + patchify: uint8 NHWC -> bf16 [B*gh*gw, p*p*3] rows, the
// A operand of the patch-embedding GEMM (vision_transformer.py:235-248 Conv2D k=s=p 'valid' is
// a pure gather for NHWC input).  One thread = 4 pixels of one image row (p % 4 == 0).
// one quad: 12 normalised values -> 24 bytes of its patch row
template <int MODE>
__device__ __forceinline__ void np_store_quad(const uint32_t (&w)[3], bf16_t* __restrict__ out, int n, int y, int x0, int P, int ps, int gh, int gw,
                                              int K, const NormConst& nc) {
    uint8_t b[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) { b[i] = (uint8_t)(w[0] >> (8 * i)); b[4 + i] = (uint8_t)(w[1] >> (8 * i)); b[8 + i] = (uint8_t)(w[2] >> (8 * i)); }
    float f[12];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (MODE == 0) {
            f[3 * p + 0] = norm1<0>(b[3 * p + 2], 0, nc);
            f[3 * p + 1] = norm1<0>(b[3 * p + 1], 1, nc);
            f[3 * p + 2] = norm1<0>(b[3 * p + 0], 2, nc);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) f[3 * p + c] = norm1<MODE>(b[3 * p + c], c, nc);
        }
    }
    int py, px, ry, rx;
    if (ps >= 0) { py = y >> ps; px = x0 >> ps; ry = y & (P - 1); rx = x0 & (P - 1); }
    else { py = y / P; px = x0 / P; ry = y - py * P; rx = x0 - px * P; }
    const int64_t row = ((int64_t)n * gh + py) * gw + px;
    const int col = (ry * P + rx) * 3;
    uint2* d = reinterpret_cast<uint2*>(out + row * K + col);          // 24 bytes, 8-byte aligned (col * 2 is a multiple of 24)
    d[0] = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
    d[1] = make_uint2(pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
    d[2] = make_uint2(pack_bf16x2(f[8], f[9]), pack_bf16x2(f[10], f[11]));
}

__device__ __forceinline__ void np_load_quad(const uint8_t* __restrict__ s, bool dw, uint32_t (&w)[3]) {
    if (dw) {
        const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s);
        w[0] = s4[0]; w[1] = s4[1]; w[2] = s4[2];
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) w[i] = (uint32_t)s[4 * i] | ((uint32_t)s[4 * i + 1] << 8) | ((uint32_t)s[4 * i + 2] << 16) | ((uint32_t)s[4 * i + 3] << 24);
    }
}

// grid = (groups of 4 * ROWS image rows, B); wave = ROWS consecutive rows, lane = 4-pixel quad.  No integer division in the loop (r03:
// the flat 64-bit quad index of rounds 1-2 cost eight divisions per quad), and TWO quads per trip: both loads are issued before either
// is converted, so a wave keeps two rows' worth of requests in flight.
template <int MODE, int ROWS>
__global__ void __launch_bounds__(256) normalize_patchify_kernel(const uint8_t* __restrict__ in, bf16_t* __restrict__ out, int B, int H,
                                                                 int W, int P, int gh, int gw, NormConst nc) {
    const int wq = (gw * P) >> 2;
    const int hh = gh * P;
    const int K = P * P * 3;
    const int n = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row0 = blockIdx.x * (4 * ROWS) + wave * ROWS;
    if (row0 >= hh) return;
    const int nrows = min(ROWS, hh - row0);
    const int ps = ((P & (P - 1)) == 0) ? (31 - __builtin_clz(P)) : -1;
    const bool dw = ((W & 3) == 0) && !((uintptr_t)in & 3);          // 12-byte quads are dword-aligned: three dword loads instead of twelve byte loads
    const uint8_t* img = in + (int64_t)n * H * W * 3;
    int k = 0, xq = lane;
    while (xq >= wq) { xq -= wq; ++k; }
    const int adv_k = 64 / wq, adv_x = 64 - adv_k * wq;
    while (k < nrows) {
        int k2 = k + adv_k, xq2 = xq + adv_x;
        if (xq2 >= wq) { xq2 -= wq; ++k2; }
        const bool two = k2 < nrows;
        uint32_t wa[3], wb[3] = {0u, 0u, 0u};
        np_load_quad(img + ((int64_t)(row0 + k) * W + xq * 4) * 3, dw, wa);
        if (two) np_load_quad(img + ((int64_t)(row0 + k2) * W + xq2 * 4) * 3, dw, wb);
        np_store_quad<MODE>(wa, out, n, row0 + k, xq * 4, P, ps, gh, gw, K, nc);
        if (two) np_store_quad<MODE>(wb, out, n, row0 + k2, xq2 * 4, P, ps, gh, gw, K, nc);
        k = k2 + adv_k;
        xq = xq2 + adv_x;
        if (xq >= wq) { xq -= wq; ++k; }
    }
}

// float32 NHWC (already normalised, the reference model's own input) -> bf16 patch rows
__global__ void __launch_bounds__(256) patchify_f32_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int B, int H, int W, int P,
                                                           int gh, int gw) {
    const int wq = (gw * P) >> 2;
    const int hh = gh * P;
    const int64_t total = (int64_t)B * hh * wq;
    const int K = P * P * 3;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int xq = (int)(q % wq);
        const int64_t r = q / wq;
        const int y = (int)(r % hh);
        const int n = (int)(r / hh);
        const int x0 = xq * 4;
        const float* s = in + (((int64_t)n * H + y) * W + x0) * 3;
        const int64_t row = ((int64_t)n * gh + y / P) * gw + x0 / P;
        const int col = ((y % P) * P + (x0 % P)) * 3;
        uint32_t* d = reinterpret_cast<uint32_t*>(out + row * K + col);
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = pack_bf16x2(s[2 * i], s[2 * i + 1]);
    }
}


// ---- per-image op dispatch (RandomChoice(elementwise=True), image_augmentations.py:563-570 + 606-617) ----------------------
// The reference maps `_random_transforms` over the batch with tf.map_fn, every image as a batch-1 tensor: op index, sign draw,
// cutout centre and Contrast's constant are PER IMAGE.  One slot of such a scheme is ONE launch here: blockIdx.y = image, the
// block reads that image's 64-byte record (op id + parameters, written by the host from the explicit decisions) and branches
// once, uniformly, into the op's body; bodies are the arithmetic of the batch kernels above, bit for bit.  The two statistics ops
// (AutoContrast, Equalize) become a per-(image, channel) 256-entry table: a histogram pass over the images that chose them, a
// table-building pass (AutoContrast's table is its own fp32 expression evaluated on the 256 possible inputs), then the same
// gather as Equalize.  RGB only (the schemes' InputSpec is uint8 NHWC and Color needs 3 channels).
struct __attribute__((aligned(16))) AugItem {
    int32_t op;
    int32_t i0, i1, i2, i3;
    int32_t pad[3];
    float f[8];
};
static_assert(sizeof(AugItem) == 64, "AugItem is the 64-byte record of include/chambers_hip.h");

template <bool FAST>
__device__ __forceinline__ void load_quad(const uint8_t* __restrict__ rowp, int x0, int W, uint8_t (&b)[12]) {
    if (FAST) {
        unpack12(*reinterpret_cast<const px4_t*>(rowp + (int64_t)x0 * 3), b);
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) b[i] = (x0 + i / 3 < W) ? rowp[(int64_t)x0 * 3 + i] : (uint8_t)0;
    }
}
template <bool FAST>
__device__ __forceinline__ void store_quad(uint8_t* __restrict__ rowp, int x0, int W, const uint8_t (&b)[12]) {
    if (FAST) {
        *reinterpret_cast<px4_t*>(rowp + (int64_t)x0 * 3) = pack12(b);
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i)
            if (x0 + i / 3 < W) rowp[(int64_t)x0 * 3 + i] = b[i];
    }
}

// histogram of the images whose slot op is AutoContrast or Equalize; grid = (slices, B)
__global__ void __launch_bounds__(256) hist_sel_kernel(const uint8_t* __restrict__ in, int32_t* __restrict__ ws, int HW,
                                                       const AugItem* __restrict__ items) {
    const int n = blockIdx.y;
    const int op = items[n].op;
    if (op != CHB_AUG_AUTOCONTRAST && op != CHB_AUG_EQUALIZE) return;   // uniform per block
    __shared__ int32_t h[3 * 256];
    for (int i = threadIdx.x; i < 768; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const uint8_t* img = in + (int64_t)n * HW * 3;
    const int64_t nbytes = (int64_t)HW * 3;
    const bool aligned = (((uintptr_t)img) & 3) == 0;
    const int64_t ng = aligned ? nbytes / 12 : 0;
    const px4_t* p4 = reinterpret_cast<const px4_t*>(img);
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += (int64_t)gridDim.x * blockDim.x) {
        uint8_t b[12];
        unpack12(p4[g], b);
#pragma unroll
        for (int i = 0; i < 12; ++i) atomicAdd(&h[(i % 3) * 256 + b[i]], 1);
    }
    for (int64_t i = ng * 12 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&h[(int)(i % 3) * 256 + img[i]], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += blockDim.x)
        if (h[i]) atomicAdd(&ws[(int64_t)n * 768 + i], h[i]);
}

// histogram -> table, in place; grid = B * 3 blocks (image, channel), thread = input value
__global__ void __launch_bounds__(256) lut_build_kernel(int32_t* __restrict__ ws, const AugItem* __restrict__ items) {
    const int n = blockIdx.x / 3;
    const int op = items[n].op;
    if (op != CHB_AUG_AUTOCONTRAST && op != CHB_AUG_EQUALIZE) return;
    __shared__ int32_t s[256];
    __shared__ int32_t first_nz, last_nz;
    int32_t* h = ws + (int64_t)blockIdx.x * 256;
    const int t = threadIdx.x;
    const int32_t mine = h[t];
    s[t] = mine;
    if (t == 0) { first_nz = 255; last_nz = 0; }
    __syncthreads();
    if (mine != 0) { atomicMax(&last_nz, t); atomicMin(&first_nz, t); }
    __syncthreads();
    int32_t lut = t;
    if (op == CHB_AUG_AUTOCONTRAST) {   // :72-86 evaluated on the value t (same fp32 sequence as autocontrast_apply_kernel)
        const float lo = (float)first_nz, hi = (float)last_nz;
        const float rng = hi - lo;
        float sc = (rng != 0.0f) ? 255.0f / rng : 0.0f;
        float of = (-lo) * sc;
        const float mask = hi > lo ? 1.0f : 0.0f;
        sc = sc * mask + (1.0f - mask);
        of = of * mask;
        float v = (float)t * sc;
        v = v + of;
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        lut = (int32_t)trunc_u8(v);
    } else {                            // tfa.image.equalize (as equalize_lut_kernel)
        for (int o = 1; o < 256; o <<= 1) {
            const int32_t v = (t >= o) ? s[t - o] : 0;
            __syncthreads();
            s[t] += v;
            __syncthreads();
        }
        const int32_t total = s[255];
        const int32_t excl = s[t] - mine;
        const int32_t step = (total - h[last_nz]) / 255;
        __syncthreads();
        if (step != 0) {
            lut = (excl + step / 2) / step;
            lut = lut < 0 ? 0 : (lut > 255 ? 255 : lut);
        }
    }
    h[t] = lut;
}

__device__ __forceinline__ uint8_t blend_rt(uint8_t deg, uint8_t x, float factor, bool clip) {
    return clip ? blend1<true>(deg, x, factor) : blend1<false>(deg, x, factor);
}

// grid = (row groups of 16, B); wave = 4 consecutive rows, lane = 4-pixel quad
template <bool FAST>
__global__ void __launch_bounds__(256) aug_dispatch_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int H, int W,
                                                           const AugItem* __restrict__ items, const int32_t* __restrict__ ws) {
    __shared__ uint8_t lut[768];
    const int n = blockIdx.y;
    const AugItem it = items[n];
    const int op = it.op;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t row_bytes = (int64_t)W * 3;
    const int64_t img_bytes = (int64_t)H * row_bytes;
    const uint8_t* img = in + (int64_t)n * img_bytes;
    uint8_t* oimg = out + (int64_t)n * img_bytes;
    const int wq = (W + 3) >> 2;
    const int row0 = blockIdx.x * 16 + wave * 4;
    if (op == CHB_AUG_AUTOCONTRAST || op == CHB_AUG_EQUALIZE) {
        for (int i = threadIdx.x; i < 768; i += blockDim.x) lut[i] = (uint8_t)ws[(int64_t)n * 768 + i];
        __syncthreads();
    }
    if (row0 >= H) return;
    const float factor = it.f[0];
    const bool clip = !(factor > 0.0f && factor < 1.0f);
    PwParams pp{0, factor, it.i0, it.i1};

    if (op == CHB_AUG_AFFINE) {   // tfa.image.transform, nearest, constant fill; projective row of every scheme op is 0
        const float a0 = it.f[0], a1 = it.f[1], a2 = it.f[2], b0 = it.f[3], b1 = it.f[4], b2 = it.f[5];
        const uint32_t fillw = ((uint32_t)(it.i0 & 0xff)) * 0x01010101u;
        const float fW = (float)W, fH = (float)H;
        for (int xq = lane; xq < wq; xq += 64) {
            const int x0 = xq * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = row0 + k;
                if (y >= H) break;
                const float fy = (float)y;
                const float ay = a1 * fy, by = b1 * fy;
                uint8_t b[12];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float fx = (float)(x0 + i);
                    const float ix = (a0 * fx + ay) + a2;
                    const float iy = (b0 * fx + by) + b2;
                    const float rx = roundf(ix), ry = roundf(iy);   // half away from zero
                    const bool ok = (rx >= 0.0f) && (rx < fW) && (ry >= 0.0f) && (ry < fH);
                    const int64_t off = ok ? ((int64_t)(int)ry * W + (int)rx) * 3 : 0;
                    uint32_t v;
                    if (FAST) {   // one unaligned dword; the last pixel of the image steps back a byte instead of reading past it
                        const int over = (off + 4 > img_bytes) ? 1 : 0;
                        v = reinterpret_cast<const u32_unaligned*>(img + off - over)->v >> (8 * over);
                    } else {
                        v = (uint32_t)img[off] | ((uint32_t)img[off + 1] << 8) | ((uint32_t)img[off + 2] << 16);
                    }
                    v = ok ? v : fillw;
                    b[3 * i + 0] = v & 0xff; b[3 * i + 1] = (v >> 8) & 0xff; b[3 * i + 2] = (v >> 16) & 0xff;
                }
                store_quad<FAST>(oimg + (int64_t)y * row_bytes, x0, W, b);
            }
        }
        return;
    }
    if (op == CHB_AUG_SHARPNESS) {   // tfa.image.sharpness: 3x3 [[1,1,1],[1,5,1],[1,1,1]]/13 on the interior, row-major fp32 sum
        const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
        const int mode = factor == 0.0f ? 0 : (clip ? 2 : 1);
        for (int xq = lane; xq < wq; xq += 64) {
            const int x0 = xq * 4;
            for (int k = 0; k < 4; ++k) {
                const int y = row0 + k;
                if (y >= H) break;
                const bool yin = (y >= 1) && (y < H - 1);
                uint8_t b[12];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int x = x0 + i;
                    const bool live = x < W;
                    const bool interior = live && yin && (x >= 1) && (x < W - 1);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const uint8_t orig = live ? img[((int64_t)y * W + x) * 3 + c] : (uint8_t)0;
                        uint8_t deg = orig;
                        if (interior) {
                            float acc = 0.0f;
#pragma unroll
                            for (int ky = -1; ky <= 1; ++ky)
#pragma unroll
                                for (int kx = -1; kx <= 1; ++kx) {
                                    const float v = (float)img[((int64_t)(y + ky) * W + (x + kx)) * 3 + c];
                                    acc = acc + v * ((ky == 0 && kx == 0) ? k5 : k1);
                                }
                            deg = trunc_u8(acc);
                        }
                        b[3 * i + c] = mode == 0 ? deg : blend_rt(deg, orig, factor, mode == 2);
                    }
                }
                store_quad<FAST>(oimg + (int64_t)y * row_bytes, x0, W, b);
            }
        }
        return;
    }
    // row-local ops: load 4 rows, transform, store
    for (int xq = lane; xq < wq; xq += 64) {
        const int x0 = xq * 4;
        uint8_t b[4][12];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (row0 + k < H) load_quad<FAST>(img + (int64_t)(row0 + k) * row_bytes, x0, W, b[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = row0 + k;
            if (y >= H) break;
            switch (op) {   // uniform
                case CHB_AUG_AUTOCONTRAST:
                case CHB_AUG_EQUALIZE:
#pragma unroll
                    for (int i = 0; i < 12; ++i) b[k][i] = lut[(i % 3) * 256 + b[k][i]];
                    break;
                case CHB_AUG_INVERT: pointwise12<CHB_PW_INVERT>(b[k], pp); break;
                case CHB_AUG_POSTERIZE: pointwise12<CHB_PW_POSTERIZE>(b[k], pp); break;
                case CHB_AUG_SOLARIZE: pointwise12<CHB_PW_SOLARIZE>(b[k], pp); break;
                case CHB_AUG_SOLARIZE_ADD: pointwise12<CHB_PW_SOLARIZE_ADD>(b[k], pp); break;
                case CHB_AUG_BRIGHTNESS: pointwise12<CHB_PW_BRIGHTNESS>(b[k], pp); break;
                case CHB_AUG_CONTRAST: pointwise12<CHB_PW_CONTRAST>(b[k], pp); break;
                case CHB_AUG_COLOR: pointwise12<CHB_PW_COLOR>(b[k], pp); break;
                case CHB_AUG_CUTOUT: {   // i0 = cy, i1 = cx, i2 = half, i3 = value
                    const int xa = max(0, it.i1 - it.i2), xb = min(W, it.i1 + it.i2);
                    const bool rowin = (y >= max(0, it.i0 - it.i2)) && (y < min(H, it.i0 + it.i2));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool inside = rowin && (x0 + i >= xa) && (x0 + i < xb);
#pragma unroll
                        for (int c = 0; c < 3; ++c) b[k][3 * i + c] = inside ? (uint8_t)it.i3 : b[k][3 * i + c];
                    }
                    break;
                }
                default: break;   // CHB_AUG_IDENTITY: RandomChance not taken
            }
            store_quad<FAST>(oimg + (int64_t)y * row_bytes, x0, W, b[k]);
        }
    }
}

// ---- fused scheme stage: RandAugment / AutoAugment (batch-shared decisions) -> [normalise -> bf16 patch rows] -------------------
// augmentation_schemes.py:204-213 (RandAugment.call) / :151-160 -> image_augmentations.py:659-665 -> vision_transformer.py:235-248.
// The unfused path runs every selected op as its own HBM round trip and then the normalise + patchify pass (n + 1 passes).  Here the
// output pixel is EVALUATED: level L of the chain is a function of level L-1 at the same pixel (pointwise ops, table ops), at
// another pixel (nearest-neighbour warp, CutOut keeps or replaces) or at the 3x3 neighbourhood (Sharpness), down to a load of the
// untouched input - the arithmetic of every level is the stand-alone kernel's, so the result is bit-identical to the op chain.  One
// read of the uint8 batch (re-reads of neighbours come from L1 / L2), one write (uint8 image, or bf16 patch rows with the "tf"
// normalisation applied); AutoContrast / Equalize at level L cost one extra pass: the histogram of level L-1 (evaluated the same
// way) and the table kernel, before the final launch.
struct FusedOp {
    int32_t op;          // CHB_AUG_*
    int32_t i0, i1, i2, i3;
    float f[6];
    int32_t pad;
};
struct FusedParams {
    int32_t n;
    FusedOp ops[CHB_FUSED_MAX_OPS];
    const int32_t* lut[CHB_FUSED_MAX_OPS];        // level's [B][3][256] table (AutoContrast / Equalize), else NULL
    const int32_t* centers[CHB_FUSED_MAX_OPS];    // level's [B][2] cutout centres (cy, cx), else NULL
    int32_t B, H, W;
    const FusedOp* items;                          // NULL, or per-image records [n][B] (elementwise schemes): they replace ops[]
    const int32_t* order;                          // NULL, or image indices sorted by launch group: blockIdx.y = position n0 + .. in it
    int32_t n0;
};

// the op record of level l for image n: the launch's own, or - elementwise schemes (image_augmentations.py:563-570), every image its
// own chain - the image's record in device memory.  A workgroup works on one image, so either way this is a scalar load.
// the image a workgroup works on: blockIdx.y, or through the group order of a sorted elementwise batch (chb_aug_fused_items)
__device__ __forceinline__ int fused_image(const FusedParams& P) { return P.order ? P.order[P.n0 + (int)blockIdx.y] : (int)blockIdx.y; }

template <bool ITEMS>
__device__ __forceinline__ const FusedOp& fused_op(const FusedParams& P, int n, int l) {
    if (ITEMS) return P.items[(int64_t)l * P.B + n];
    return P.ops[l];
}

// one unaligned dword per pixel (the memory pipeline, not HBM, bounds the gathers: a byte load costs what a dword load costs);
// the last pixel of the image steps back a byte instead of reading past the allocation (a one-pixel image loads bytes)
__device__ __forceinline__ uint32_t px_load(const uint8_t* __restrict__ img, int H, int W, int y, int x) {
    const int off = (y * W + x) * 3;
    const int end = H * W * 3;
    if (off + 4 <= end) return reinterpret_cast<const u32_unaligned*>(img + off)->v & 0xffffffu;
    if (off >= 1) return reinterpret_cast<const u32_unaligned*>(img + off - 1)->v >> 8;
    return (uint32_t)img[off] | ((uint32_t)img[off + 1] << 8) | ((uint32_t)img[off + 2] << 16);
}

struct FusedCtx {
    const uint8_t* img;      // this image, level -1
    const uint8_t* lut;      // LDS: [level][3][256] tables of the table ops of this segment
    int n;                   // image index (cutout centres)
    bool fast;               // W % 4 == 0 and 4-byte aligned rows: 12-byte quad loads
};

__device__ __forceinline__ uint32_t px_pointwise(int op, uint32_t v, const FusedOp& o) {
    uint8_t c[3] = {(uint8_t)(v & 0xff), (uint8_t)((v >> 8) & 0xff), (uint8_t)((v >> 16) & 0xff)};
    const float factor = o.f[0];
    const bool clip = !(factor > 0.0f && factor < 1.0f);
    switch (op) {
        case CHB_AUG_INVERT:
            for (int k = 0; k < 3; ++k) c[k] = 255 - c[k];
            break;
        case CHB_AUG_POSTERIZE:
            for (int k = 0; k < 3; ++k) c[k] = (uint8_t)((c[k] >> o.i0) << o.i0);
            break;
        case CHB_AUG_SOLARIZE:
            for (int k = 0; k < 3; ++k) c[k] = ((int)c[k] < o.i0) ? c[k] : (uint8_t)(255 - c[k]);
            break;
        case CHB_AUG_SOLARIZE_ADD:
            for (int k = 0; k < 3; ++k) {
                int t = (int)c[k] + o.i1;
                t = t < 0 ? 0 : (t > 255 ? 255 : t);
                c[k] = ((int)c[k] < o.i0) ? (uint8_t)t : c[k];
            }
            break;
        case CHB_AUG_BRIGHTNESS:
            for (int k = 0; k < 3; ++k) c[k] = blend_rt(0, c[k], factor, clip);
            break;
        case CHB_AUG_CONTRAST:
            for (int k = 0; k < 3; ++k) c[k] = blend_rt((uint8_t)o.i0, c[k], factor, clip);
            break;
        case CHB_AUG_COLOR: {
            const uint8_t d = gray_u8(c[0], c[1], c[2]);
            for (int k = 0; k < 3; ++k) c[k] = blend_rt(d, c[k], factor, clip);
            break;
        }
        default: break;
    }
    return (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
}

__device__ __forceinline__ void quad_pointwise(int op, uint8_t (&b)[12], const FusedOp& o) {
    const PwParams pp{0, o.f[0], o.i0, o.i1};
    switch (op) {   // uniform over the launch
        case CHB_AUG_INVERT: pointwise12<CHB_PW_INVERT>(b, pp); break;
        case CHB_AUG_POSTERIZE: pointwise12<CHB_PW_POSTERIZE>(b, pp); break;
        case CHB_AUG_SOLARIZE: pointwise12<CHB_PW_SOLARIZE>(b, pp); break;
        case CHB_AUG_SOLARIZE_ADD: pointwise12<CHB_PW_SOLARIZE_ADD>(b, pp); break;
        case CHB_AUG_BRIGHTNESS: pointwise12<CHB_PW_BRIGHTNESS>(b, pp); break;
        case CHB_AUG_CONTRAST: pointwise12<CHB_PW_CONTRAST>(b, pp); break;
        case CHB_AUG_COLOR: pointwise12<CHB_PW_COLOR>(b, pp); break;
        default: break;
    }
}

__device__ __forceinline__ bool affine_source(const FusedOp& o, int W, int H, int x, int y, int& sx, int& sy) {
    const float fx = (float)x, fy = (float)y;
    const float ix = (o.f[0] * fx + o.f[1] * fy) + o.f[2];
    const float iy = (o.f[3] * fx + o.f[4] * fy) + o.f[5];
    const float rx = roundf(ix), ry = roundf(iy);   // half away from zero
    const bool ok = (rx >= 0.0f) && (rx < (float)W) && (ry >= 0.0f) && (ry < (float)H);
    sx = ok ? (int)rx : 0;
    sy = ok ? (int)ry : 0;
    return ok;
}

// The constant arm of a per-byte select (CutOut's value, a warp's fill), held in a vector register.  With the op records
// loop-invariant - in scalar registers - hipcc 7.2 emitted wrong byte selects for CutOut -> Equalize chains
// (profiles/r03_augment_stage.txt note (h)); behind this opaque move the same build is bit-exact.  One v_mov per level.
__device__ __forceinline__ uint32_t vgpr_byte(int v) {
    uint32_t r = (uint32_t)v & 0xffu;
#ifndef CHB_NO_VGPR_BYTE          /* tools/check_byte_select_isa.py builds the unguarded variant to show what the guard removes */
    asm volatile("" : "+v"(r));
#endif
    return r;
}

template <bool ITEMS = false>
__device__ __forceinline__ bool cutout_inside(const FusedParams& P, int l, int n, int y, int x) {
    if (ITEMS && !P.centers[l]) return false;        // per-image records on the device: the host could not check that a centre table came with them
    const int cy = P.centers[l][2 * n], cx = P.centers[l][2 * n + 1], half = fused_op<ITEMS>(P, n, l).i2;
    return (y >= max(0, cy - half)) && (y < min(P.H, cy + half)) && (x >= max(0, cx - half)) && (x < min(P.W, cx + half));
}

// ---- one pixel of level L (gathers: below a warp, and the two edge columns of a Sharpness window) ----
template <int L, bool ITEMS = false>
struct FusedEval {
    static __device__ uint32_t at(const FusedParams& P, const FusedCtx& C, int y, int x) {
        const FusedOp& o = fused_op<ITEMS>(P, C.n, L);
        const int op = o.op;     // the same for every thread of the launch
        if (op == CHB_AUG_SHARPNESS) {
            const bool interior = (y >= 1) && (y < P.H - 1) && (x >= 1) && (x < P.W - 1);
            const float factor = o.f[0];
            const bool clip = !(factor > 0.0f && factor < 1.0f);
            const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
            float acc[3] = {0.0f, 0.0f, 0.0f};
            uint32_t centre = 0;
#pragma unroll 1
            for (int t = interior ? 0 : 4; t < (interior ? 9 : 5); ++t) {      // row-major taps; a border pixel reads its centre only
                const int ky = t / 3 - 1, kx = t % 3 - 1;
                const uint32_t v = FusedEval<L - 1, ITEMS>::at(P, C, y + ky, x + kx);
                const float w = (t == 4) ? k5 : k1;
                if (t == 4) centre = v;
                acc[0] = acc[0] + (float)(v & 0xff) * w;
                acc[1] = acc[1] + (float)((v >> 8) & 0xff) * w;
                acc[2] = acc[2] + (float)((v >> 16) & 0xff) * w;
            }
            uint32_t deg = centre;
            if (interior) deg = (uint32_t)trunc_u8(acc[0]) | ((uint32_t)trunc_u8(acc[1]) << 8) | ((uint32_t)trunc_u8(acc[2]) << 16);
            if (factor == 0.0f) return deg;
            uint32_t out = 0;
            for (int k = 0; k < 3; ++k)
                out |= (uint32_t)blend_rt((uint8_t)((deg >> (8 * k)) & 0xff), (uint8_t)((centre >> (8 * k)) & 0xff), factor, clip) << (8 * k);
            return out;
        }
        int sx = x, sy = y;
        bool keep = true;            // false: the level's constant replaces the pixel
        uint32_t konst = 0;
        if (op == CHB_AUG_AFFINE) {
            keep = affine_source(o, P.W, P.H, x, y, sx, sy);
            konst = vgpr_byte(o.i0) * 0x010101u;
        } else if (op == CHB_AUG_CUTOUT) {
            keep = !cutout_inside<ITEMS>(P, L, C.n, y, x);
            konst = vgpr_byte(o.i3) * 0x010101u;
        }
        const uint32_t v = FusedEval<L - 1, ITEMS>::at(P, C, sy, sx);
        if (!keep) return konst;
        if (op == CHB_AUG_AUTOCONTRAST || op == CHB_AUG_EQUALIZE) {
            const uint8_t* lut = C.lut + L * 768;
            return (uint32_t)lut[v & 0xff] | ((uint32_t)lut[256 + ((v >> 8) & 0xff)] << 8) | ((uint32_t)lut[512 + ((v >> 16) & 0xff)] << 16);
        }
        return px_pointwise(op, v, o);
    }
};
template <bool ITEMS>
struct FusedEval<-1, ITEMS> {
    static __device__ __forceinline__ uint32_t at(const FusedParams& P, const FusedCtx& C, int y, int x) {
        return px_load(C.img, P.H, P.W, y, x);
    }
};

// ---- four pixels of level L at four arbitrary positions (below a warp): the positions travel down the chain, the values come back
// up as one 12-byte quad, so every pointwise / table level is the vector code of the quad path instead of one byte-wise evaluation
// per pixel (r03: a pixel-local op UNDER a warp cost 30 us more than the same op above it).  Sharpness under a warp keeps the
// per-pixel evaluation (nine taps around each of four unrelated positions).
template <int L, bool ITEMS = false>
struct FusedGather {
    static __device__ __forceinline__ void at(const FusedParams& P, const FusedCtx& C, const int (&ys)[4], const int (&xs)[4], uint8_t (&b)[12]) {
        const FusedOp& o = fused_op<ITEMS>(P, C.n, L);
        const int op = o.op;
        if (op == CHB_AUG_SHARPNESS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t v = FusedEval<L, ITEMS>::at(P, C, ys[i], xs[i]);
                b[3 * i + 0] = v & 0xff; b[3 * i + 1] = (v >> 8) & 0xff; b[3 * i + 2] = (v >> 16) & 0xff;
            }
            return;
        }
        if (op == CHB_AUG_AFFINE) {
            int y2[4], x2[4];
            bool ok[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ok[i] = affine_source(o, P.W, P.H, xs[i], ys[i], x2[i], y2[i]);
            FusedGather<L - 1, ITEMS>::at(P, C, y2, x2, b);
            const uint8_t fill = (uint8_t)vgpr_byte(o.i0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) b[3 * i + c] = ok[i] ? b[3 * i + c] : fill;
            return;
        }
        FusedGather<L - 1, ITEMS>::at(P, C, ys, xs, b);
        if (op == CHB_AUG_AUTOCONTRAST || op == CHB_AUG_EQUALIZE) {
            const uint8_t* lut = C.lut + L * 768;
#pragma unroll
            for (int i = 0; i < 12; ++i) b[i] = lut[(i % 3) * 256 + b[i]];
        } else if (op == CHB_AUG_CUTOUT) {
            const uint8_t cutv = (uint8_t)vgpr_byte(o.i3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool inside = cutout_inside<ITEMS>(P, L, C.n, ys[i], xs[i]);
#pragma unroll
                for (int c = 0; c < 3; ++c) b[3 * i + c] = inside ? cutv : b[3 * i + c];
            }
        } else {
            quad_pointwise(op, b, o);
        }
    }
};
template <bool ITEMS>
struct FusedGather<-1, ITEMS> {
    static __device__ __forceinline__ void at(const FusedParams& P, const FusedCtx& C, const int (&ys)[4], const int (&xs)[4], uint8_t (&b)[12]) {
#pragma unroll      // the four gathers in flight together
        for (int i = 0; i < 4; ++i) {
            const uint32_t v = px_load(C.img, P.H, P.W, ys[i], xs[i]);
            b[3 * i + 0] = v & 0xff; b[3 * i + 1] = (v >> 8) & 0xff; b[3 * i + 2] = (v >> 16) & 0xff;
        }
    }
};

// ---- four consecutive pixels (x0 % 4 == 0) of row y at level L; pixels at x >= W come back as anything and are never used ----
// MODE (decided by the host per launch, so that a launch carries only the code it needs - a third of the registers and twice the
// waves in flight for the lean ones):
//   FUSED_LOCAL   no warp, no Sharpness: a level reads only its own pixel;
//   FUSED_ROWS    no Sharpness, and every warp is ROW-CONTIGUOUS (TranslateX / TranslateY / ShearX; affine_row_contiguous checked every
//                 pixel of this H x W): the source of (x, y) is (x + d(y), r(y)), so a quad reads ONE run of four pixels - the
//                 levels underneath are evaluated as a quad at the shifted start (any x0, also outside the row: masked on the way
//                 up), one affine evaluation per quad instead of four and no per-pixel gathers;
//   FUSED_GENERAL everything else (per-pixel gathers below a warp, windows below a Sharpness);
//   FUSED_ITEMS   as FUSED_GENERAL, the op records being each image's own (FusedParams::items: elementwise schemes).
//   FUSED_ITEMS_LOCAL / FUSED_ITEMS_ROWS: the lean modes over per-image records - the images of an elementwise batch are sorted by
//                 what their chain needs and every group gets its own launch (chb_aug_fused_items with an order).
#ifndef CHB_GENERAL_TILES
#define CHB_GENERAL_TILES 1          // A/B: 0 = the row-long mapping in the general launches too
#endif
constexpr int FUSED_GENERAL = 0, FUSED_LOCAL = 1, FUSED_ROWS = 2, FUSED_ITEMS = 3, FUSED_ITEMS_LOCAL = 5, FUSED_ITEMS_ROWS = 6;
constexpr bool mode_items(int m) { return m == FUSED_ITEMS || m == FUSED_ITEMS_LOCAL || m == FUSED_ITEMS_ROWS; }
constexpr int mode_shape(int m) { return m == FUSED_ITEMS_LOCAL ? FUSED_LOCAL : (m == FUSED_ITEMS_ROWS ? FUSED_ROWS : (m == FUSED_ITEMS ? FUSED_GENERAL : m)); }
struct __attribute__((aligned(4))) u32x4_a4 { uint32_t w[4]; };

template <int L, int MODE>
struct FusedQuad {
    static __device__ __forceinline__ void at(const FusedParams& P, const FusedCtx& C, int y, int x0, uint8_t (&b)[12]) {
        constexpr bool ITEMS = mode_items(MODE);               // per-image records
        constexpr bool GENERAL = mode_shape(MODE) == FUSED_GENERAL;
        const FusedOp& o = fused_op<ITEMS>(P, C.n, L);
        const int op = o.op;
        if (GENERAL && op == CHB_AUG_AFFINE) {
            int ys[4], xs[4];
            bool ok[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ok[i] = affine_source(o, P.W, P.H, x0 + i, y, xs[i], ys[i]);
            FusedGather<L - 1, ITEMS>::at(P, C, ys, xs, b);
            const uint8_t fill = (uint8_t)vgpr_byte(o.i0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) b[3 * i + c] = ok[i] ? b[3 * i + c] : fill;
            return;
        }
        if (GENERAL && op == CHB_AUG_SHARPNESS) {
            // window rows y-1..y+1, columns x0-1..x0+4: three quads + the two edge columns, 18 evaluations for 4 outputs; the
            // taps are summed in row-major order, so the rows can be folded into the 12 sums as they arrive
            const bool yin = (y >= 1) && (y < P.H - 1);
            const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
            float acc[12];
            uint8_t orig[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) { acc[i] = 0.0f; orig[i] = 0; }
#pragma unroll 1
            for (int r = yin ? 0 : 1; r < (yin ? 3 : 2); ++r) {
                uint8_t w[18];
                uint8_t q[12];
                FusedQuad<L - 1, MODE>::at(P, C, y + r - 1, x0, q);
                const uint32_t lft = (yin && x0 >= 1) ? FusedEval<L - 1, ITEMS>::at(P, C, y + r - 1, x0 - 1) : 0u;
                const uint32_t rgt = (yin && x0 + 4 < P.W) ? FusedEval<L - 1, ITEMS>::at(P, C, y + r - 1, x0 + 4) : 0u;
                w[0] = lft & 0xff; w[1] = (lft >> 8) & 0xff; w[2] = (lft >> 16) & 0xff;
#pragma unroll
                for (int i = 0; i < 12; ++i) w[3 + i] = q[i];
                w[15] = rgt & 0xff; w[16] = (rgt >> 8) & 0xff; w[17] = (rgt >> 16) & 0xff;
                const float kmid = (r == 1) ? k5 : k1;
                if (r == 1) {
#pragma unroll
                    for (int i = 0; i < 12; ++i) orig[i] = q[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) acc[3 * i + c] = acc[3 * i + c] + (float)w[3 * (i + kx) + c] * (kx == 1 ? kmid : k1);
            }
            const float factor = o.f[0];
            const bool clip = !(factor > 0.0f && factor < 1.0f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int x = x0 + i;
                const bool interior = yin && (x >= 1) && (x < P.W - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint8_t deg = interior ? trunc_u8(acc[3 * i + c]) : orig[3 * i + c];
                    b[3 * i + c] = (factor == 0.0f) ? deg : blend_rt(deg, orig[3 * i + c], factor, clip);
                }
            }
            return;
        }
        FusedQuad<L - 1, MODE>::at(P, C, y, x0, b);
        if (op == CHB_AUG_AUTOCONTRAST || op == CHB_AUG_EQUALIZE) {
            const uint8_t* lut = C.lut + L * 768;
#pragma unroll
            for (int i = 0; i < 12; ++i) b[i] = lut[(i % 3) * 256 + b[i]];
        } else if (op == CHB_AUG_CUTOUT) {
            const uint8_t cutv = (uint8_t)vgpr_byte(o.i3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool inside = cutout_inside<ITEMS>(P, L, C.n, y, x0 + i);
#pragma unroll
                for (int c = 0; c < 3; ++c) b[3 * i + c] = inside ? cutv : b[3 * i + c];
            }
        } else {
            quad_pointwise(op, b, o);
        }
    }
};
template <int MODE>
struct FusedQuad<-1, MODE> {
    static __device__ __forceinline__ void at(const FusedParams& P, const FusedCtx& C, int y, int x0, uint8_t (&b)[12]) {
        const uint8_t* row = C.img + (int64_t)y * P.W * 3;
        if (C.fast) load_quad<true>(row, x0, P.W, b);
        else load_quad<false>(row, x0, P.W, b);
    }
};

// ---- FUSED_ROWS: four pixels of ONE row y of level L at columns xs[0..3] (a run x0 .. x0+3 at the top; under a warp whatever its
// row map makes of them - for a pure shift again a run).  Level -1 loads a run with one 16-byte access, anything else per pixel.
template <int L, bool ITEMS = false>
struct FusedRow {
    static __device__ __forceinline__ void at(const FusedParams& P, const FusedCtx& C, int y, const int (&xs)[4], uint8_t (&b)[12]) {
        const FusedOp& o = fused_op<ITEMS>(P, C.n, L);
        const int op = o.op;
        if (op == CHB_AUG_AFFINE) {             // f[3] == 0 (host): the source row does not depend on x
            const float fy = (float)y;
            const float ry = roundf((o.f[3] * 0.0f + o.f[4] * fy) + o.f[5]);
            const bool rowok = (ry >= 0.0f) && (ry < (float)P.H);
            const uint8_t fill = (uint8_t)vgpr_byte(o.i0);
            int x2[4];
            bool ok[4];
            if (o.pad == 1) {                   // a pure shift on this H x W (affine_row_contiguous): no per-pixel evaluation at all
                const int d = (int)roundf((o.f[0] * 0.0f + o.f[1] * fy) + o.f[2]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { x2[i] = xs[i] + d; ok[i] = rowok && (x2[i] >= 0) && (x2[i] < P.W); }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float rx = roundf((o.f[0] * (float)xs[i] + o.f[1] * fy) + o.f[2]);
                    ok[i] = rowok && (rx >= 0.0f) && (rx < (float)P.W);
                    x2[i] = ok[i] ? (int)rx : 0;
                }
            }
            if (!(ok[0] || ok[1] || ok[2] || ok[3])) {
#pragma unroll
                for (int i = 0; i < 12; ++i) b[i] = fill;
                return;
            }
            FusedRow<L - 1, ITEMS>::at(P, C, (int)ry, x2, b);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) b[3 * i + c] = ok[i] ? b[3 * i + c] : fill;
            return;
        }
        FusedRow<L - 1, ITEMS>::at(P, C, y, xs, b);
        if (op == CHB_AUG_AUTOCONTRAST || op == CHB_AUG_EQUALIZE) {
            const uint8_t* lut = C.lut + L * 768;
#pragma unroll
            for (int i = 0; i < 12; ++i) b[i] = lut[(i % 3) * 256 + b[i]];
        } else if (op == CHB_AUG_CUTOUT) {
            const uint8_t cutv = (uint8_t)vgpr_byte(o.i3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool inside = cutout_inside<ITEMS>(P, L, C.n, y, xs[i]);
#pragma unroll
                for (int c = 0; c < 3; ++c) b[3 * i + c] = inside ? cutv : b[3 * i + c];
            }
        } else {
            quad_pointwise(op, b, o);
        }
    }
};
template <bool ITEMS>
struct FusedRow<-1, ITEMS> {
    static __device__ __forceinline__ void at(const FusedParams& P, const FusedCtx& C, int y, const int (&xs)[4], uint8_t (&b)[12]) {
        // A run (also one that leaves the row: those pixels are masked above): one 16-byte load from the dword below + byte
        // alignment.  Bytes in front of / behind this image belong to its neighbours in the batch; only the first and the last
        // image have none there.  Everything else: per-pixel loads at columns clamped into the row.
        const int off = (y * P.W + xs[0]) * 3, lo = off & ~3, end = P.H * P.W * 3;
        const bool run = (xs[1] == xs[0] + 1) && (xs[2] == xs[0] + 2) && (xs[3] == xs[0] + 3);
        if (run && (lo >= 0 || C.n > 0) && (lo + 16 <= end || C.n + 1 < P.B)) {
            const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(C.img + lo);
            const uint32_t sh = (uint32_t)(off & 3);
            const uint32_t w0 = __builtin_amdgcn_alignbyte(v.w[1], v.w[0], sh), w1 = __builtin_amdgcn_alignbyte(v.w[2], v.w[1], sh),
                           w2 = __builtin_amdgcn_alignbyte(v.w[3], v.w[2], sh);
#pragma unroll
            for (int k = 0; k < 4; ++k) { b[k] = (w0 >> (8 * k)) & 0xff; b[4 + k] = (w1 >> (8 * k)) & 0xff; b[8 + k] = (w2 >> (8 * k)) & 0xff; }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t v = px_load(C.img, P.H, P.W, y, min(max(xs[i], 0), P.W - 1));
                b[3 * i + 0] = v & 0xff; b[3 * i + 1] = (v >> 8) & 0xff; b[3 * i + 2] = (v >> 16) & 0xff;
            }
        }
    }
};

// the quad (y, x0 .. x0+3) of the launch's top level
template <int L, int MODE>
__device__ __forceinline__ void fused_top_quad(const FusedParams& P, const FusedCtx& C, int y, int x0, uint8_t (&b)[12]) {
    if (mode_shape(MODE) == FUSED_ROWS) {
        const int xs[4] = {x0, x0 + 1, x0 + 2, x0 + 3};
        FusedRow<L, mode_items(MODE)>::at(P, C, y, xs, b);
    } else {
        FusedQuad<L, MODE>::at(P, C, y, x0, b);
    }
}

template <bool ITEMS = false>
__device__ __forceinline__ void fused_stage_luts(const FusedParams& P, int n, uint8_t* lutS) {
    for (int l = 0; l < P.n; ++l)       // uniform
        if (P.lut[l] && (fused_op<ITEMS>(P, n, l).op == CHB_AUG_AUTOCONTRAST || fused_op<ITEMS>(P, n, l).op == CHB_AUG_EQUALIZE))
            for (int i = threadIdx.x; i < 768; i += blockDim.x) lutS[l * 768 + i] = (uint8_t)P.lut[l][(int64_t)n * 768 + i];
    __syncthreads();
}

// histogram of level NLEV-1 (the input of the table op at level NLEV); grid = (row slices, B).  Every workgroup leaves its OWN
// partial table in `part` ([B][slices][768], plain stores: nothing to zero beforehand, no global atomics - r03: the zeroing launch in
// front of this one cost a launch gap per table op); fused_lut_kernel adds the slices up.  The LDS bins are swizzled (a Posterize in
// front leaves multiples of 8: one bank in eight).  `npop` "popular" values the host expects from the ops underneath - 0 and 255 behind
// an op that clips (Brightness, Contrast, Color, SolarizeAdd), the fill value behind a warp - are counted by ballot into scalar
// registers and skipped by the atomics: a same-address LDS atomic of 64 lanes is 64 serial updates (r02: X>Equalize 200-230 us
// behind such an X, 106 us behind Invert).
__device__ __forceinline__ int hist_slot(int bin) { return bin ^ ((bin >> 3) & 7) ^ (((bin >> 6) & 3) << 3); }

template <int NLEV, int MODE>
__global__ void __launch_bounds__(256) fused_hist_kernel(const uint8_t* __restrict__ in, int32_t* __restrict__ part, FusedParams P, int fast,
                                                         int minmax, int npop, int pop0, int pop1, int pop2) {
    if (mode_items(MODE)) {         // per-image chains: this image's op at the level - a table op at all, and which
        const int op = fused_op<true>(P, fused_image(P), NLEV < CHB_FUSED_MAX_OPS ? NLEV : 0).op;
        if (op != CHB_AUG_AUTOCONTRAST && op != CHB_AUG_EQUALIZE) return;
        minmax = op == CHB_AUG_AUTOCONTRAST ? 1 : 0;
    }
    __shared__ int32_t h[768];
    int lo[3] = {255, 255, 255}, hi[3] = {0, 0, 0};
    __shared__ uint8_t lutS[CHB_FUSED_MAX_OPS * 768];
    for (int i = threadIdx.x; i < 768; i += blockDim.x) h[i] = 0;
    const int n = fused_image(P);
    fused_stage_luts<mode_items(MODE)>(P, n, lutS);
    const FusedCtx C{in + (int64_t)n * P.H * P.W * 3, lutS, n, fast != 0};
    const int wq = (P.W + 3) >> 2;
    const int nq = P.H * wq;
    // (row, quad) advance by the grid stride without a division per quad: one division per thread up front
    const int stride = gridDim.x * blockDim.x;
    const int adv_y = stride / wq, adv_x = stride - adv_y * wq;
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    int y = q / wq, xq = q - y * wq;
    const int pops[3] = {pop0, pop1, pop2};
    int cnt[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};        // [popular value][channel], wave-uniform (scalar registers)
    // general chains (per-pixel gathers): 16 x 16 pixel tiles per wave as in fused_final_kernel, the image's tiles dealt round the
    // slice's waves; lanes of a tile that hangs over the image stay in the loop (the ballots below) and count nothing
    constexpr bool TILED = CHB_GENERAL_TILES && mode_shape(MODE) == FUSED_GENERAL && NLEV > 0;
    const int tiles_x = (wq + 3) >> 2, tiles = tiles_x * ((P.H + 15) >> 4);
    const int lane = threadIdx.x & 63;
    int t = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int t_stride = gridDim.x * (blockDim.x >> 6);
    for (; TILED ? t < tiles : q < nq; q += stride, t += t_stride) {
        bool valid = true;
        if (TILED) {
            const int ty = t / tiles_x;
            y = ty * 16 + (lane >> 2);
            xq = (t - ty * tiles_x) * 4 + (lane & 3);
            valid = (y < P.H) && (xq < wq);
        }
        const int x0 = xq * 4;
        uint8_t b[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) b[i] = 0;
        if (valid) fused_top_quad<NLEV - 1, MODE>(P, C, y, x0, b);
        if (!TILED) {
            y += adv_y;
            xq += adv_x;
            if (xq >= wq) { xq -= wq; ++y; }
        }
        if (minmax) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (valid && x0 + i / 3 < P.W) { lo[i % 3] = min(lo[i % 3], (int)b[i]); hi[i % 3] = max(hi[i % 3], (int)b[i]); }
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int v = (valid && x0 + i / 3 < P.W) ? (int)b[i] : -1;
                bool rare = v >= 0;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (k < npop) {         // uniform
                        cnt[k][i % 3] += (int)__builtin_popcountll(__ballot(v == pops[k]));
                        rare = rare && (v != pops[k]);
                    }
                if (rare) atomicAdd(&h[(i % 3) * 256 + hist_slot(v)], 1);
            }
        }
    }
    // the partial tables are indexed by the image, or - a sorted elementwise batch - by its position in the level's order
    int32_t* mine = part + ((int64_t)(P.order ? P.n0 + (int)blockIdx.y : n) * gridDim.x + blockIdx.x) * 768;
    if (minmax) {      // AutoContrast needs the extremes only: slots 0 / 1 of each channel's partial table hold min / max
        __syncthreads();        // (the zeroing above is complete)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            for (int o = 32; o > 0; o >>= 1) {
                lo[c] = min(lo[c], __shfl_xor(lo[c], o));
                hi[c] = max(hi[c], __shfl_xor(hi[c], o));
            }
            if ((threadIdx.x & 63) == 0) { h[(threadIdx.x >> 6) * 8 + 2 * c] = lo[c]; h[(threadIdx.x >> 6) * 8 + 2 * c + 1] = hi[c]; }
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            const int c = threadIdx.x;
            mine[c * 256] = min(min(h[2 * c], h[8 + 2 * c]), min(h[16 + 2 * c], h[24 + 2 * c]));
            mine[c * 256 + 1] = max(max(h[2 * c + 1], h[8 + 2 * c + 1]), max(h[16 + 2 * c + 1], h[24 + 2 * c + 1]));
        }
        return;
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (k < npop && cnt[k][c]) atomicAdd(&h[c * 256 + hist_slot(pops[k])], cnt[k][c]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += blockDim.x) mine[i] = h[(i & ~255) + hist_slot(i & 255)];
}

// final pass: grid = (row groups of 4 RW, B); wave = RW consecutive rows, lane = 4-pixel quad.
// PATCH: "tf" normalisation + bf16 patch rows (patch % 4 == 0: a quad never straddles patches), else uint8 NHWC
template <int NLEV, bool PATCH, int MODE>
__global__ void __launch_bounds__(256) fused_final_kernel(const uint8_t* __restrict__ in, void* __restrict__ out, FusedParams P, int patch, int gh, int gw,
                                                          int fast) {
    __shared__ uint8_t lutS[CHB_FUSED_MAX_OPS * 768];
    const int n = fused_image(P);
    fused_stage_luts<mode_items(MODE)>(P, n, lutS);
    const FusedCtx C{in + (int64_t)n * P.H * P.W * 3, lutS, n, fast != 0};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wq = PATCH ? (gw * patch) >> 2 : (P.W + 3) >> 2;
    const int hh = PATCH ? gh * patch : P.H;
    // rows per wave: 8 where a quad is cheap (FUSED_ROWS, like fused_local_kernel: with 4 the per-workgroup set-up showed), 4 under the
    // per-pixel gathers of the general modes (8 measured 10 % slower there)
    constexpr int RW = (mode_shape(MODE) == FUSED_ROWS) ? 8 : 4;
    const int row0 = blockIdx.x * (4 * RW) + wave * RW;
    const int K = patch * patch * 3;
    // The wave's RW rows x wq quads as one index space: all 64 lanes busy whatever the row length.  No integer division in the loop
    // (r03: five of them per quad - idx / wq and the patch coordinates - cost more vector instructions than a pixel-local chain
    // itself): (row, quad) advance by 64 quads per trip, and a power-of-two patch edge (16, 32: every ViT of the zoo) turns the patch
    // coordinates into shifts and masks; other edges keep the division.
    const int ps = (PATCH && (patch & (patch - 1)) == 0) ? (31 - __builtin_clz(patch)) : -1;     // uniform
    auto quad = [&](int y, int x0) {
        uint8_t b[12];
        fused_top_quad<NLEV - 1, MODE>(P, C, y, x0, b);
        if (PATCH) {
            float f[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) f[i] = norm1<1>(b[i], i % 3, NormConst{});
            int py, px, ry, rx;
            if (ps >= 0) { py = y >> ps; px = x0 >> ps; ry = y & (patch - 1); rx = x0 & (patch - 1); }
            else { py = y / patch; px = x0 / patch; ry = y - py * patch; rx = x0 - px * patch; }
            const int64_t row = ((int64_t)n * gh + py) * gw + px;
            const int col = (ry * patch + rx) * 3;
            uint2* d = reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + row * K + col);     // 24 bytes, 8-byte aligned (col * 2 is a multiple of 24)
            d[0] = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
            d[1] = make_uint2(pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
            d[2] = make_uint2(pack_bf16x2(f[8], f[9]), pack_bf16x2(f[10], f[11]));
        } else {
            uint8_t* orow = reinterpret_cast<uint8_t*>(out) + ((int64_t)n * P.H + y) * P.W * 3;
            if (C.fast) store_quad<true>(orow, x0, P.W, b);
            else store_quad<false>(orow, x0, P.W, b);
        }
    };
    if (CHB_GENERAL_TILES && mode_shape(MODE) == FUSED_GENERAL) {
        // General chains gather per pixel: a wave takes 16 x 16 pixel TILES of the workgroup's 16-row band (lane = row of the tile x
        // quad), so that one gather instruction covers a compact block - under a rotation its source is a rotated block of ~22 x 22
        // pixels, some thirty cache lines, where the 256 x 1 strip of the row-long mapping crosses more than a hundred - and with
        // 16-pixel patches the wave writes one whole patch row (1536 contiguous bytes).
        const int y = blockIdx.x * 16 + (lane >> 2);
        for (int t = wave; t * 4 < wq; t += 4) {
            const int xq = t * 4 + (lane & 3);
            if (y < hh && xq < wq) quad(y, xq * 4);
        }
        return;
    }
    if (row0 >= hh) return;
    const int nrows = min(RW, hh - row0);
    int k = 0, xq = lane;
    while (xq >= wq) { xq -= wq; ++k; }
    const int adv_k = 64 / wq, adv_x = 64 - adv_k * wq;       // 64 quads = adv_k whole rows + adv_x quads (uniform, once per wave)
    for (; k < nrows; ) {
        quad(row0 + k, xq * 4);
        k += adv_k;
        xq += adv_x;
        if (xq >= wq) { xq -= wq; ++k; }
    }
}

// ---- FUSED_LOCAL chains, final pass: every level reads only its own pixel, so the launch is the normalise + patchify pass with the
// chain applied to the quad between load and store - and it takes that kernel's shape (r03): 8 rows per wave, TWO quads per trip with
// both loads issued before either chain runs.  grid = (groups of 32 rows, B).
template <int L, bool ITEMS = false>
struct FusedApply {
    static __device__ __forceinline__ void on(const FusedParams& P, const FusedCtx& C, int y, int x0, uint8_t (&b)[12]) {
        FusedApply<L - 1, ITEMS>::on(P, C, y, x0, b);
        const FusedOp& o = fused_op<ITEMS>(P, C.n, L);
        const int op = o.op;
        if (op == CHB_AUG_AUTOCONTRAST || op == CHB_AUG_EQUALIZE) {
            const uint8_t* lut = C.lut + L * 768;
#pragma unroll
            for (int i = 0; i < 12; ++i) b[i] = lut[(i % 3) * 256 + b[i]];
        } else if (op == CHB_AUG_CUTOUT) {
            const uint8_t cutv = (uint8_t)vgpr_byte(o.i3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool inside = cutout_inside<ITEMS>(P, L, C.n, y, x0 + i);
#pragma unroll
                for (int c = 0; c < 3; ++c) b[3 * i + c] = inside ? cutv : b[3 * i + c];
            }
        } else {
            quad_pointwise(op, b, o);
        }
    }
};
template <bool ITEMS>
struct FusedApply<-1, ITEMS> {
    static __device__ __forceinline__ void on(const FusedParams&, const FusedCtx&, int, int, uint8_t (&)[12]) {}
};

template <int NLEV, bool PATCH, bool ITEMS = false>
__global__ void __launch_bounds__(256) fused_local_kernel(const uint8_t* __restrict__ in, void* __restrict__ out, FusedParams P, int patch, int gh, int gw,
                                                          int fast) {
    constexpr int ROWS = 8;
    __shared__ uint8_t lutS[CHB_FUSED_MAX_OPS * 768];
    const int n = fused_image(P);
    fused_stage_luts<ITEMS>(P, n, lutS);
    const FusedCtx C{in + (int64_t)n * P.H * P.W * 3, lutS, n, fast != 0};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wq = PATCH ? (gw * patch) >> 2 : (P.W + 3) >> 2;
    const int hh = PATCH ? gh * patch : P.H;
    const int row0 = blockIdx.x * (4 * ROWS) + wave * ROWS;
    const int K = patch * patch * 3;
    if (row0 >= hh) return;
    const int nrows = min(ROWS, hh - row0);
    const int ps = (PATCH && (patch & (patch - 1)) == 0) ? (31 - __builtin_clz(patch)) : -1;     // uniform
    auto emit = [&](int y, int x0, uint8_t (&b)[12]) {
        if (PATCH) {
            float f[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) f[i] = norm1<1>(b[i], i % 3, NormConst{});
            int py, px, ry, rx;
            if (ps >= 0) { py = y >> ps; px = x0 >> ps; ry = y & (patch - 1); rx = x0 & (patch - 1); }
            else { py = y / patch; px = x0 / patch; ry = y - py * patch; rx = x0 - px * patch; }
            const int64_t row = ((int64_t)n * gh + py) * gw + px;
            const int col = (ry * patch + rx) * 3;
            uint2* d = reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + row * K + col);     // 24 bytes, 8-byte aligned
            d[0] = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
            d[1] = make_uint2(pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
            d[2] = make_uint2(pack_bf16x2(f[8], f[9]), pack_bf16x2(f[10], f[11]));
        } else {
            uint8_t* orow = reinterpret_cast<uint8_t*>(out) + ((int64_t)n * P.H + y) * P.W * 3;
            if (C.fast) store_quad<true>(orow, x0, P.W, b);
            else store_quad<false>(orow, x0, P.W, b);
        }
    };
    int k = 0, xq = lane;
    while (xq >= wq) { xq -= wq; ++k; }
    const int adv_k = 64 / wq, adv_x = 64 - adv_k * wq;
    while (k < nrows) {
        int k2 = k + adv_k, xq2 = xq + adv_x;
        if (xq2 >= wq) { xq2 -= wq; ++k2; }
        const bool two = k2 < nrows;
        uint8_t ba[12], bb[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) bb[i] = 0;
        FusedQuad<-1, FUSED_LOCAL>::at(P, C, row0 + k, xq * 4, ba);
        if (two) FusedQuad<-1, FUSED_LOCAL>::at(P, C, row0 + k2, xq2 * 4, bb);
        FusedApply<NLEV - 1, ITEMS>::on(P, C, row0 + k, xq * 4, ba);
        emit(row0 + k, xq * 4, ba);
        if (two) {
            FusedApply<NLEV - 1, ITEMS>::on(P, C, row0 + k2, xq2 * 4, bb);
            emit(row0 + k2, xq2 * 4, bb);
        }
        k = k2 + adv_k;
        xq = xq2 + adv_x;
        if (xq >= wq) { xq -= wq; ++k; }
    }
}

// ---- a Sharpness with the chain around it in ONE launch (r03; before: the levels below into a uint8 scratch image, the stand-alone
// Sharpness into another, then normalise + patchify - three launches and two round trips).  A wave walks down 8 output rows, a lane
// owns the same quad in all of them; each input row is a quad of level S-1, evaluated once where it is needed (BMODE: the launch
// mode of the levels below; S = 0: plain loads), the halo pixels come from the neighbouring lanes.  Every input byte is converted and
// multiplied by 1/13 once, the per-output sum keeps the reference's row-major order (as in sharpness_rows_kernel).  The levels above
// the Sharpness must be pixel-local (host): they are applied to the finished quad, then the quad leaves as uint8 or as
// "tf"-normalised bf16 patch rows.  grid = (groups of 4 x 8 rows, B).
template <int S, int BMODE>
__device__ __forceinline__ void fused_below_quad(const FusedParams& P, const FusedCtx& C, int y, int x0, uint8_t (&q)[12]) {
    if (S == 0) FusedQuad<-1, FUSED_LOCAL>::at(P, C, y, x0, q);
    else fused_top_quad<(S > 0 ? S - 1 : 0), BMODE>(P, C, y, x0, q);
}

template <int S, bool PATCH, int BMODE>
__global__ void __launch_bounds__(256) fused_sharp_kernel(const uint8_t* __restrict__ in, void* __restrict__ out, FusedParams P, int patch, int gh, int gw,
                                                          int fast) {
    constexpr int R = 8;             // output rows per wave: R + 2 rows of level S-1 are evaluated for them (1.25 per output row)
    __shared__ uint8_t lutS[CHB_FUSED_MAX_OPS * 768];
    const int n = blockIdx.y;
    fused_stage_luts(P, n, lutS);
    const FusedCtx C{in + (int64_t)n * P.H * P.W * 3, lutS, n, fast != 0};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int H = P.H, W = P.W;
    const int wq = PATCH ? (gw * patch) >> 2 : (W + 3) >> 2;       // quads written per row
    const int wq_in = (W + 3) >> 2;                                 // quads that exist in a row (halo source)
    const int hh = PATCH ? gh * patch : H;
    const int y0 = (blockIdx.x * 4 + wave) * R;
    if (y0 >= hh) return;
    const int y1 = min(y0 + R, hh);                                 // output rows y0 .. y1-1
    const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
    const FusedOp& so = P.ops[S];
    const float factor = so.f[0];
    const bool clip = !(factor > 0.0f && factor < 1.0f);
    const int K = patch * patch * 3;
    const int ps = (PATCH && (patch & (patch - 1)) == 0) ? (31 - __builtin_clz(patch)) : -1;     // uniform
    // a window of 64 lanes = 62 output quads + one halo quad on either side (evaluated, not written): every halo pixel is a
    // neighbouring lane's.  The wave walks DOWN its rows with a three-row window of products: one evaluation of the chain below per
    // input row, one copy of its code in the kernel.
    for (int xb = 0; xb < wq; xb += 62) {
        const int xq = xb - 1 + lane;
        const bool live = (xq >= 0) && (xq < wq_in);
        const bool writes = (lane >= 1) && (lane <= 62) && (xq < wq);
        const int x0 = xq * 4;
        float pa[18], pb[18];     // (float)byte * k1 for [left px | own 4 px | right px] of input rows y-1, y
        uint8_t ownb[12];         // level S-1 at row y
#pragma unroll
        for (int i = 0; i < 18; ++i) { pa[i] = 0.0f; pb[i] = 0.0f; }
#pragma unroll
        for (int i = 0; i < 12; ++i) ownb[i] = 0;
#pragma unroll 1
        for (int yy = y0 - 1; yy <= y1; ++yy) {            // input row yy completes the window of output row yy - 1
            const bool rin = (yy >= 0) && (yy < H);        // rows outside the image are never used: border rows keep the original
            uint8_t q[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) q[i] = 0;
            if (rin && live) fused_below_quad<S, BMODE>(P, C, yy, x0, q);
            const uint32_t lft = __shfl_up((uint32_t)q[9] | ((uint32_t)q[10] << 8) | ((uint32_t)q[11] << 16), 1, 64);
            const uint32_t rgt = __shfl_down((uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16), 1, 64);
            float pc[18];
            pc[0] = (float)(lft & 0xff) * k1;
            pc[1] = (float)((lft >> 8) & 0xff) * k1;
            pc[2] = (float)((lft >> 16) & 0xff) * k1;
#pragma unroll
            for (int i = 0; i < 12; ++i) pc[3 + i] = (float)q[i] * k1;
            pc[15] = (float)(rgt & 0xff) * k1;
            pc[16] = (float)((rgt >> 8) & 0xff) * k1;
            pc[17] = (float)((rgt >> 16) & 0xff) * k1;
            const int y = yy - 1;
            if (y >= y0 && writes) {                       // (y < y1 by the loop bound)
                const bool yin = (y >= 1) && (y < H - 1);
                uint8_t b[12];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int x = x0 + i;
                    const bool interior = yin && (x >= 1) && (x < W - 1);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const uint8_t orig = ownb[3 * i + c];
                        // window column (i + kx) of a row is pixel x - 1 + kx; pa, pb, pc are rows y-1, y, y+1
                        float acc = pa[3 * i + c];
                        acc = acc + pa[3 * (i + 1) + c];
                        acc = acc + pa[3 * (i + 2) + c];
                        acc = acc + pb[3 * i + c];
                        acc = acc + (float)orig * k5;
                        acc = acc + pb[3 * (i + 2) + c];
                        acc = acc + pc[3 * i + c];
                        acc = acc + pc[3 * (i + 1) + c];
                        acc = acc + pc[3 * (i + 2) + c];
                        const uint8_t deg = interior ? trunc_u8(acc) : orig;
                        b[3 * i + c] = (factor == 0.0f) ? deg : blend_rt(deg, orig, factor, clip);
                    }
                }
                for (int l = S + 1; l < P.n; ++l) {          // the pixel-local levels above (uniform)
                    const FusedOp& o = P.ops[l];
                    if (o.op == CHB_AUG_AUTOCONTRAST || o.op == CHB_AUG_EQUALIZE) {
                        const uint8_t* lut = C.lut + l * 768;
#pragma unroll
                        for (int i = 0; i < 12; ++i) b[i] = lut[(i % 3) * 256 + b[i]];
                    } else if (o.op == CHB_AUG_CUTOUT) {
                        const uint8_t cutv = (uint8_t)vgpr_byte(o.i3);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool inside = cutout_inside(P, l, C.n, y, x0 + i);
#pragma unroll
                            for (int c = 0; c < 3; ++c) b[3 * i + c] = inside ? cutv : b[3 * i + c];
                        }
                    } else {
                        quad_pointwise(o.op, b, o);
                    }
                }
                if (PATCH) {
                    float f[12];
#pragma unroll
                    for (int i = 0; i < 12; ++i) f[i] = norm1<1>(b[i], i % 3, NormConst{});
                    int py, px, ry, rx;
                    if (ps >= 0) { py = y >> ps; px = x0 >> ps; ry = y & (patch - 1); rx = x0 & (patch - 1); }
                    else { py = y / patch; px = x0 / patch; ry = y - py * patch; rx = x0 - px * patch; }
                    const int64_t row = ((int64_t)n * gh + py) * gw + px;
                    const int col = (ry * patch + rx) * 3;
                    uint2* d = reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + row * K + col);     // 24 bytes, 8-byte aligned
                    d[0] = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
                    d[1] = make_uint2(pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
                    d[2] = make_uint2(pack_bf16x2(f[8], f[9]), pack_bf16x2(f[10], f[11]));
                } else {
                    uint8_t* orow = reinterpret_cast<uint8_t*>(out) + ((int64_t)n * H + y) * W * 3;
                    if (C.fast) store_quad<true>(orow, x0, W, b);
                    else store_quad<false>(orow, x0, W, b);
                }
            }
#pragma unroll
            for (int i = 0; i < 18; ++i) { pa[i] = pb[i]; pb[i] = pc[i]; }
#pragma unroll
            for (int i = 0; i < 12; ++i) ownb[i] = q[i];
        }
    }
}

// table from the histogram of one (image, channel): the batch-shared op of level `lvl` (same code as lut_build_kernel); the histogram
// is the sum of the `slices` partial tables fused_hist_kernel left for the image (AutoContrast: min / max over their slots 0 / 1)
// (items != NULL: the image's own op at the level, [B] records; an image without a table op there is skipped)
__global__ void __launch_bounds__(256) fused_lut_kernel(int32_t* __restrict__ tables, const int32_t* __restrict__ part, int slices, int op,
                                                        const FusedOp* __restrict__ items = nullptr, const int32_t* __restrict__ order = nullptr) {
    __shared__ int32_t s[256];
    __shared__ int32_t hsave[256];
    // order != NULL (sorted elementwise batch): workgroup triple `pos` belongs to image order[pos], its partial tables sit at pos
    const int pos = blockIdx.x / 3, c = blockIdx.x - 3 * pos;
    const int n = order ? order[pos] : pos;
    if (items) {
        op = items[n].op;
        if (op != CHB_AUG_AUTOCONTRAST && op != CHB_AUG_EQUALIZE) return;
    }
    const int t = threadIdx.x;
    const int32_t* p0 = part + (int64_t)pos * slices * 768 + c * 256 + t;
    int32_t mine = 0;
    if (op == CHB_AUG_AUTOCONTRAST) {
        if (t < 2) {
            mine = p0[0];
            for (int k = 1; k < slices; ++k) mine = (t == 0) ? min(mine, p0[(int64_t)k * 768]) : max(mine, p0[(int64_t)k * 768]);
        }
    } else {
#pragma unroll 4
        for (int k = 0; k < slices; ++k) mine += p0[(int64_t)k * 768];
    }
    int32_t lut = t;
    if (op == CHB_AUG_AUTOCONTRAST) {
        s[t] = mine;
        __syncthreads();
        const float lo = (float)s[0], hi = (float)s[1];      // the statistics pass left min / max in slots 0 / 1
        const float rng = hi - lo;
        float sc = (rng != 0.0f) ? 255.0f / rng : 0.0f;
        float of = (-lo) * sc;
        const float mask = hi > lo ? 1.0f : 0.0f;
        sc = sc * mask + (1.0f - mask);
        of = of * mask;
        float v = (float)t * sc;
        v = v + of;
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        lut = (int32_t)trunc_u8(v);
    } else {
        // inclusive sum over the 256 bins: shuffles inside each wave, the four wave totals through LDS (one barrier; r03: the
        // 8-step LDS scan with its 16 barriers was most of this launch's 13 us); the last occupied bin from the waves' ballots
        const int lane = t & 63, w = t >> 6;
        int32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        const unsigned long long nz = __ballot(mine != 0);
        if (lane == 63) s[w] = incl;
        if (lane == 0) s[4 + w] = nz ? (w * 64 + 63 - __builtin_clzll(nz)) : 0;
        hsave[t] = mine;
        __syncthreads();
        int32_t base = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) base += (k < w) ? s[k] : 0;
        incl += base;
        const int32_t total = s[0] + s[1] + s[2] + s[3];
        const int32_t last = max(max(s[4], s[5]), max(s[6], s[7]));        // 0 for an empty histogram, as before
        const int32_t excl = incl - mine;
        const int32_t step = (total - hsave[last]) / 255;
        if (step != 0) {
            lut = (excl + step / 2) / step;
            lut = lut < 0 ? 0 : (lut > 255 ? 255 : lut);
        }
    }
    tables[((int64_t)n * 3 + c) * 256 + t] = lut;
}

const NormConst kCaffe = {{103.939f, 116.779f, 123.68f}, {1.f, 1.f, 1.f}};
const NormConst kTorch = {{0.485f, 0.456f, 0.406f}, {0.229f, 0.224f, 0.225f}};

}  // namespace

extern "C" {

int chb_aug_pointwise(const uint8_t* in, uint8_t* out, int64_t n_bytes, int op, float factor, int i0, int i1, void* stream) {
    if (n_bytes < 0) return CHB_EINVAL;
    if (n_bytes == 0) return CHB_OK;
    if (!in || !out) return CHB_EINVAL;
    if (((uintptr_t)in & 3) || ((uintptr_t)out & 3)) return CHB_EINVAL;
    if (op == CHB_PW_COLOR && (n_bytes % 3) != 0) return CHB_EINVAL;
    PwParams pp{op, factor, i0, i1};
    const int64_t ng = n_bytes / 12;
    const int grid = stream_grid(ng);
    hipStream_t s = (hipStream_t)stream;
#define CHB_PW_CASE(OP) \
    case OP: hipLaunchKernelGGL(pointwise_kernel<OP>, dim3(grid), dim3(256), 0, s, in, out, ng, n_bytes, pp); break;
    switch (op) {
        CHB_PW_CASE(CHB_PW_INVERT)
        CHB_PW_CASE(CHB_PW_POSTERIZE)
        CHB_PW_CASE(CHB_PW_SOLARIZE)
        CHB_PW_CASE(CHB_PW_SOLARIZE_ADD)
        CHB_PW_CASE(CHB_PW_BRIGHTNESS)
        CHB_PW_CASE(CHB_PW_CONTRAST)
        CHB_PW_CASE(CHB_PW_COLOR)
        default: return CHB_EINVAL;
    }
#undef CHB_PW_CASE
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_aug_affine(const uint8_t* in, uint8_t* out, int B, int H, int W, int C, const float* transform_host8,
                   const float* transforms_dev, int per_image, int fill, void* stream) {
    if (B == 0) return CHB_OK;
    if (!in || !out || B < 0 || H <= 0 || W <= 0 || C <= 0 || C > 4) return CHB_EINVAL;
    if (!transform_host8 && !transforms_dev) return CHB_EINVAL;
    float t[8] = {1, 0, 0, 0, 1, 0, 0, 0};
    if (transform_host8) for (int i = 0; i < 8; ++i) t[i] = transform_host8[i];
    if (transform_host8 && t[6] == 0.0f && t[7] == 0.0f && C == 3 && (W & 3) == 0 && !((uintptr_t)out & 3) &&
        (int64_t)H * W * 3 < 2147483647LL) {
        // rows stay rows (shear, translate): the row-per-wave kernel; rows mix (rotation): the tiled one.  CHB_AFFINE_ALGO forces
        // 1 = rows, 2 = 32 x 8 tiles, 3 = 16 x 16 tiles (A/B timing).
        int algo = chb_option(CHB_OPT_AFFINE_ALGO);
        if (algo < 1 || algo > 3) algo = (t[3] == 0.0f) ? 1 : 3;
        if (algo == 1) {
            hipLaunchKernelGGL(affine_rgb_kernel, dim3(row_grid(((int64_t)B * H + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W,
                               t[0], t[1], t[2], t[3], t[4], t[5], fill & 0xff);
        } else {
            const int TW = algo == 2 ? 32 : 16, TH = 256 / TW;
            const int64_t tiles = (int64_t)B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
            int64_t blocks = (tiles + 3) / 4;
            if (blocks > 32768) blocks = 32768;
            auto kern = algo == 2 ? affine_rgb_tile_kernel<32> : affine_rgb_tile_kernel<16>;
            hipLaunchKernelGGL(kern, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, t[0], t[1], t[2], t[3], t[4],
                               t[5], fill & 0xff);
        }
        CHB_LAUNCH_CHECK();
        return CHB_OK;
    }
    hipLaunchKernelGGL(affine_kernel, dim3(row_grid(((int64_t)B * H + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, C,
                       transform_host8 ? nullptr : transforms_dev, per_image, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7],
                       fill & 0xff);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_aug_cutout(const uint8_t* in, uint8_t* out, int B, int H, int W, int C, const int32_t* centers_dev, int mask_size,
                   int value, void* stream) {
    if (B == 0) return CHB_OK;
    if (!in || !out || !centers_dev || B < 0 || H <= 0 || W <= 0 || C <= 0 || C > 4) return CHB_EINVAL;
    if (mask_size < 0 || (mask_size & 1)) return CHB_EINVAL;  // tfa: mask_size must be even
    hipLaunchKernelGGL(cutout_kernel, dim3(row_grid(((int64_t)B * H + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, C,
                       centers_dev, mask_size / 2, value & 0xff);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

static int slices_for(int64_t bytes_per_image, int B) {
    // enough blocks to fill 256 CUs x 8 even for small batches
    int64_t per = (bytes_per_image / 12 + 255) / 256;
    int64_t want = (2048 + B - 1) / (B > 0 ? B : 1);
    if (want < 1) want = 1;
    if (per < 1) per = 1;
    return (int)(want < per ? want : per);
}

int chb_aug_autocontrast(const uint8_t* in, uint8_t* out, int B, int H, int W, int C, int32_t* workspace, void* stream) {
    if (B == 0) return CHB_OK;
    if (!in || !out || !workspace || B < 0 || H <= 0 || W <= 0 || C <= 0 || C > 4) return CHB_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nws = B * C * 2;
    hipLaunchKernelGGL(stats_init_kernel, dim3(chb_div_up(nws, 256)), dim3(256), 0, s, workspace, nws, 0);
    const int sl = slices_for((int64_t)H * W * C, B);
    hipLaunchKernelGGL(minmax_kernel, dim3(sl, B), dim3(256), 0, s, in, workspace, H * W, C);
    hipLaunchKernelGGL(autocontrast_apply_kernel, dim3(sl, B), dim3(256), 0, s, in, out, workspace, H * W, C);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_aug_equalize(const uint8_t* in, uint8_t* out, int B, int H, int W, int C, int32_t* workspace, void* stream) {
    if (B == 0) return CHB_OK;
    if (!in || !out || !workspace || B < 0 || H <= 0 || W <= 0 || C <= 0 || C > 4) return CHB_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nws = B * C * 256;
    hipLaunchKernelGGL(stats_init_kernel, dim3(chb_div_up(nws, 256)), dim3(256), 0, s, workspace, nws, 1);
    const int sl = slices_for((int64_t)H * W * C, B);
    hipLaunchKernelGGL(hist_kernel, dim3(sl, B), dim3(256), 0, s, in, workspace, H * W, C);
    hipLaunchKernelGGL(equalize_lut_kernel, dim3(B * C), dim3(256), 0, s, workspace);
    hipLaunchKernelGGL(lut_apply_kernel, dim3(sl, B), dim3(256), 0, s, in, out, workspace, H * W, C);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_aug_dispatch(const uint8_t* in, uint8_t* out, int B, int H, int W, const void* items_dev, int n_stats, int32_t* workspace,
                     void* stream) {
    if (B == 0) return CHB_OK;
    if (!in || !out || !items_dev || B < 0 || H <= 0 || W <= 0 || n_stats < 0) return CHB_EINVAL;
    if (n_stats > 0 && !workspace) return CHB_EINVAL;
    if ((int64_t)H * W * 3 >= 2147483647LL || B > 65535) return CHB_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const AugItem* items = (const AugItem*)items_dev;
    if (n_stats > 0) {
        const int nws = B * 768;
        hipLaunchKernelGGL(stats_init_kernel, dim3(chb_div_up(nws, 256)), dim3(256), 0, s, workspace, nws, 1);
        const int sl = slices_for((int64_t)H * W * 3, n_stats);
        hipLaunchKernelGGL(hist_sel_kernel, dim3(sl, B), dim3(256), 0, s, in, workspace, H * W, items);
        hipLaunchKernelGGL(lut_build_kernel, dim3(B * 3), dim3(256), 0, s, workspace, items);
    }
    const dim3 grid((H + 15) / 16, B);
    const bool fast = (W & 3) == 0 && !((uintptr_t)in & 3) && !((uintptr_t)out & 3);
    if (fast) hipLaunchKernelGGL(aug_dispatch_kernel<true>, grid, dim3(256), 0, s, in, out, B, H, W, items, workspace);
    else hipLaunchKernelGGL(aug_dispatch_kernel<false>, grid, dim3(256), 0, s, in, out, B, H, W, items, workspace);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

static bool fused_is_local(int op) { return op != CHB_AUG_AFFINE && op != CHB_AUG_SHARPNESS; }

// Is the nearest-neighbour warp `o` on an H x W image a pure per-row shift - source of (x, y) = (x + d(y), r(y)) for EVERY x of the row,
// d(y) the rounded source column of x = 0?  (Not so where a source column is an exact tie: half-away rounding flips with the sign and
// with the binade of the sum - ShearX by -0.27 on rows 50 and 150 of a 224-row image.)  Decided by evaluating the device's own expression (same float order, this file is built without contraction) at
// every pixel, once per (coefficients, H, W): RandAugment draws a sign, not a magnitude, so a run sees two records per op.
// `budget` (optional): evaluations this caller may still pay for - a per-image sort of records with free-running coefficients would
// otherwise walk H x W pixels per image; past the budget the answer is "no" (always correct: the per-pixel path), uncached.
static bool affine_row_contiguous(const FusedOp& o, int H, int W, int* budget = nullptr) {
    if (o.f[0] != 1.0f || o.f[3] != 0.0f) return false;
    struct Entry { float f[6]; int H, W; bool yes; };
    static std::mutex mu;
    static std::vector<Entry> seen;
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry& e : seen)
        if (e.H == H && e.W == W && !memcmp(e.f, o.f, sizeof(e.f))) return e.yes;
    if (budget && (*budget)-- <= 0) return false;
    bool yes = true;
    for (int y = 0; y < H && yes; ++y) {
        const float fy = (float)y;
        const float d = roundf((o.f[0] * 0.0f + o.f[1] * fy) + o.f[2]);
        if (!(fabsf(d) < 1.0e6f)) { yes = false; break; }
        for (int x = 0; x < W; ++x) {
            const float fx = (float)x;
            if (roundf((o.f[0] * fx + o.f[1] * fy) + o.f[2]) != fx + d) { yes = false; break; }
        }
    }
    if (seen.size() >= 256) seen.clear();
    Entry e;
    memcpy(e.f, o.f, sizeof(e.f));
    e.H = H; e.W = W; e.yes = yes;
    seen.push_back(e);
    return yes;
}

// workspace of ONE table op of chb_aug_fused: its [B][3][256] table, then the [B][slices][768] partial histograms it is made from
static int64_t fused_table_ints(int B, int H, int W) { return (int64_t)B * 768 * (1 + slices_for((int64_t)H * W * 3, B)); }

int64_t chb_aug_fused_workspace_ints(int B, int H, int W, int n_tables) {
    if (B <= 0 || H <= 0 || W <= 0 || n_tables <= 0) return 0;
    return (int64_t)n_tables * fused_table_ints(B, H, W);
}

// one segment of a chain: ops[0..n) evaluated per output pixel of `src`; returns the number of tables it used
// `sharp` >= 0: ops[sharp] is a Sharpness whose levels above are all pixel-local - the final launch is fused_sharp_kernel
static int fused_segment(const uint8_t* src, void* dst, int B, int H, int W, int n_ops, const FusedOp* ops, const int32_t* const* centers,
                         int32_t* ws, int patch, hipStream_t s, int sharp = -1) {
    FusedParams P;
    memset(&P, 0, sizeof(P));
    P.n = n_ops; P.B = B; P.H = H; P.W = W;
    const int fast = ((W & 3) == 0 && !((uintptr_t)src & 3) && (patch || !((uintptr_t)dst & 3))) ? 1 : 0;
    int n_tables = 0;
    bool local = true, rows = true;      // rows: no Sharpness and every warp row-contiguous (FUSED_ROWS)
    for (int l = 0; l < n_ops; ++l) {
        P.ops[l] = ops[l];
        P.centers[l] = centers[l];
        local = local && fused_is_local(ops[l].op);
        rows = rows && ops[l].op != CHB_AUG_SHARPNESS && (ops[l].op != CHB_AUG_AFFINE || ops[l].f[3] == 0.0f);
        P.ops[l].pad = (ops[l].op == CHB_AUG_AFFINE && ops[l].f[3] == 0.0f && affine_row_contiguous(ops[l], H, W)) ? 1 : 0;
        if (ops[l].op == CHB_AUG_AUTOCONTRAST || ops[l].op == CHB_AUG_EQUALIZE) P.lut[l] = ws + (int64_t)(n_tables++) * fused_table_ints(B, H, W);
    }
    // table ops, in chain order: partial histograms of the level below (evaluated through everything under it), then the table
    const int slices = slices_for((int64_t)H * W * 3, B);
    for (int l = 0; l < n_ops; ++l) {
        if (!P.lut[l]) continue;
        int32_t* t = const_cast<int32_t*>(P.lut[l]);
        int32_t* part = t + (int64_t)B * 768;
        const int minmax = ops[l].op == CHB_AUG_AUTOCONTRAST ? 1 : 0;
        const dim3 grid(slices, B);
        // values the levels underneath make popular (see fused_hist_kernel): at most 0, 255 and one fill value
        int pops[3] = {-2, -2, -2}, npop = 0;
        bool clips = false;
        int fillv = -1;
        for (int k = 0; k < l; ++k) {
            const int o = ops[k].op;
            clips = clips || o == CHB_AUG_BRIGHTNESS || o == CHB_AUG_CONTRAST || o == CHB_AUG_COLOR || o == CHB_AUG_SOLARIZE_ADD;
            if (o == CHB_AUG_AFFINE) fillv = ops[k].i0 & 0xff;
            if (o == CHB_AUG_AUTOCONTRAST || o == CHB_AUG_EQUALIZE || o == CHB_AUG_INVERT || o == CHB_AUG_SOLARIZE || o == CHB_AUG_POSTERIZE) fillv = -1;   // remapped above the warp
        }
        if (clips) { pops[npop++] = 0; pops[npop++] = 255; }
        if (fillv >= 0 && !(clips && (fillv == 0 || fillv == 255))) pops[npop++] = fillv;
        bool local_l = true, rows_l = true;        // launch mode of the levels under this table op
        for (int k = 0; k < l; ++k) {
            local_l = local_l && fused_is_local(ops[k].op);
            rows_l = rows_l && ops[k].op != CHB_AUG_SHARPNESS && (ops[k].op != CHB_AUG_AFFINE || ops[k].f[3] == 0.0f);
        }
        const int np_ = minmax ? 0 : npop;
#define CHB_FUSED_HIST(NL)                                                                                                                                  \
    do {                                                                                                                                                    \
        if (local_l) hipLaunchKernelGGL((fused_hist_kernel<NL, FUSED_LOCAL>), grid, dim3(256), 0, s, src, part, P, fast, minmax, np_, pops[0], pops[1], pops[2]);   \
        else if (rows_l) hipLaunchKernelGGL((fused_hist_kernel<NL, FUSED_ROWS>), grid, dim3(256), 0, s, src, part, P, fast, minmax, np_, pops[0], pops[1], pops[2]); \
        else hipLaunchKernelGGL((fused_hist_kernel<NL, FUSED_GENERAL>), grid, dim3(256), 0, s, src, part, P, fast, minmax, np_, pops[0], pops[1], pops[2]);       \
    } while (0)
        switch (l) {
            case 0: CHB_FUSED_HIST(0); break;
            case 1: CHB_FUSED_HIST(1); break;
            case 2: CHB_FUSED_HIST(2); break;
            default: CHB_FUSED_HIST(3); break;
        }
#undef CHB_FUSED_HIST
        hipLaunchKernelGGL(fused_lut_kernel, dim3(B * 3), dim3(256), 0, s, t, part, slices, ops[l].op, (const FusedOp*)nullptr);
    }
    int gh = 0, gw = 0;
    if (patch) { gh = H / patch; gw = W / patch; }
    if (sharp >= 0) {
        bool local_b = true, rows_b = true;         // launch mode of the levels under the Sharpness
        for (int k = 0; k < sharp; ++k) {
            local_b = local_b && fused_is_local(ops[k].op);
            rows_b = rows_b && ops[k].op != CHB_AUG_SHARPNESS && (ops[k].op != CHB_AUG_AFFINE || ops[k].f[3] == 0.0f);
        }
        const int hh = patch ? gh * patch : H;
        const dim3 sgrid(((hh + 7) / 8 + 3) / 4, B);      // 8 output rows per wave (fused_sharp_kernel's R), 4 waves per workgroup
#define CHB_FUSED_SHARP2(S_, PT)                                                                                                             \
    do {                                                                                                                                     \
        if (local_b) hipLaunchKernelGGL((fused_sharp_kernel<S_, PT, FUSED_LOCAL>), sgrid, dim3(256), 0, s, src, dst, P, patch, gh, gw, fast);    \
        else if (rows_b) hipLaunchKernelGGL((fused_sharp_kernel<S_, PT, FUSED_ROWS>), sgrid, dim3(256), 0, s, src, dst, P, patch, gh, gw, fast); \
        else hipLaunchKernelGGL((fused_sharp_kernel<S_, PT, FUSED_GENERAL>), sgrid, dim3(256), 0, s, src, dst, P, patch, gh, gw, fast);        \
    } while (0)
#define CHB_FUSED_SHARP(S_)                    \
    do {                                       \
        if (patch) CHB_FUSED_SHARP2(S_, true); \
        else CHB_FUSED_SHARP2(S_, false);      \
    } while (0)
        switch (sharp) {
            case 0: CHB_FUSED_SHARP(0); break;
            case 1: CHB_FUSED_SHARP(1); break;
            case 2: CHB_FUSED_SHARP(2); break;
            default: CHB_FUSED_SHARP(3); break;
        }
#undef CHB_FUSED_SHARP
#undef CHB_FUSED_SHARP2
        return n_tables;
    }
    const dim3 lgrid(((patch ? gh * patch : H) + 31) / 32, B);      // fused_local_kernel, FUSED_ROWS: 8 rows per wave
    const dim3 grid(((patch ? gh * patch : H) + 15) / 16, B);       // general modes: 4
#define CHB_FUSED_FINAL2(NL, PT)                                                                                                            \
    do {                                                                                                                                    \
        if (local) hipLaunchKernelGGL((fused_local_kernel<NL, PT>), lgrid, dim3(256), 0, s, src, dst, P, patch, gh, gw, fast);                \
        else if (rows) hipLaunchKernelGGL((fused_final_kernel<NL, PT, FUSED_ROWS>), lgrid, dim3(256), 0, s, src, dst, P, patch, gh, gw, fast); \
        else hipLaunchKernelGGL((fused_final_kernel<NL, PT, FUSED_GENERAL>), grid, dim3(256), 0, s, src, dst, P, patch, gh, gw, fast);         \
    } while (0)
#define CHB_FUSED_FINAL(NL)                \
    do {                                   \
        if (patch) CHB_FUSED_FINAL2(NL, true); \
        else CHB_FUSED_FINAL2(NL, false);      \
    } while (0)
    switch (n_ops) {
        case 1: CHB_FUSED_FINAL(1); break;
        case 2: CHB_FUSED_FINAL(2); break;
        case 3: CHB_FUSED_FINAL(3); break;
        default: CHB_FUSED_FINAL(4); break;
    }
#undef CHB_FUSED_FINAL
#undef CHB_FUSED_FINAL2
    return n_tables;
}

int chb_aug_fused(const uint8_t* in, void* out, int B, int H, int W, int n_ops, const void* ops_host, const int32_t* const* centers_dev,
                  int32_t* workspace, uint8_t* scratch, int patch, void* stream) {
    if (B == 0) return CHB_OK;
    if (!in || !out || !ops_host || B < 0 || H <= 0 || W <= 0 || n_ops < 1 || n_ops > CHB_FUSED_MAX_OPS || patch < 0 || (patch & 3)) return CHB_EINVAL;
    if ((int64_t)H * W * 3 >= 2147483647LL - 4 || B > 65535) return CHB_EUNSUPPORTED;
    if (patch && (H / patch == 0 || W / patch == 0)) return CHB_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    FusedOp ops[CHB_FUSED_MAX_OPS];
    const int32_t* centers[CHB_FUSED_MAX_OPS];
    memcpy(ops, ops_host, sizeof(FusedOp) * n_ops);
    for (int l = 0; l < n_ops; ++l) {
        const int op = ops[l].op;
        if (op < CHB_AUG_IDENTITY || op > CHB_AUG_CUTOUT) return CHB_EINVAL;
        centers[l] = (op == CHB_AUG_CUTOUT && centers_dev) ? centers_dev[l] : nullptr;
        if (op == CHB_AUG_CUTOUT && !centers[l]) return CHB_EINVAL;
        if ((op == CHB_AUG_AUTOCONTRAST || op == CHB_AUG_EQUALIZE) && !workspace) return CHB_EINVAL;
    }
    // A Sharpness takes the levels directly under it (whatever they are) and the pixel-local levels directly above it into ITS launch
    // (fused_sharp_kernel): `lo..end` is one launch, the Sharpness at `sh`.  A table op above stays out: its histogram pass would
    // evaluate nine-tap windows per pixel, where the cut hands it a materialised image.
    auto sharp_launch = [&](int lo, int& sh, int& end) {
        sh = lo;
        while (sh < n_ops && ops[sh].op != CHB_AUG_SHARPNESS) ++sh;
        if (sh == n_ops) { end = n_ops; return false; }
        end = sh + 1;
        while (end < n_ops && fused_is_local(ops[end].op) && ops[end].op != CHB_AUG_AUTOCONTRAST && ops[end].op != CHB_AUG_EQUALIZE) ++end;
        return true;
    };
    if (!scratch) {    // no memory for an intermediate image: ONE launch
        int sh, end;
        if (sharp_launch(0, sh, end) && end == n_ops) {       // a single Sharpness with nothing but pixel-local levels above it
            fused_segment(in, out, B, H, W, n_ops, ops, centers, workspace, patch, s, sh);
        } else {                                              // the whole chain per output pixel, gathers, nine-tap windows and all
            fused_segment(in, out, B, H, W, n_ops, ops, centers, workspace, patch, s);
        }
        CHB_LAUNCH_CHECK();
        return CHB_OK;
    }
    // With scratch a chain that does not fit one launch (a warp or a second Sharpness ABOVE a Sharpness) is cut behind the launch of
    // each Sharpness: what follows reads a materialised uint8 image instead of evaluating nine-tap windows under its gathers.
    const int64_t img_bytes = (int64_t)B * H * W * 3;
    const uint8_t* src = in;
    int lo = 0, n_cut = 0;
    int32_t* ws = workspace;
    while (lo < n_ops) {
        int sh, end;
        const bool sharp = sharp_launch(lo, sh, end);
        const bool last = end == n_ops;
        void* dst = last ? out : (void*)(scratch + (int64_t)(n_cut & 1) * img_bytes);
        const int nt = fused_segment(src, dst, B, H, W, end - lo, ops + lo, centers + lo, ws, last ? patch : 0, s, sharp ? sh - lo : -1);
        if (ws) ws += (int64_t)nt * fused_table_ints(B, H, W);
        src = (const uint8_t*)dst;
        ++n_cut;
        lo = end;
    }
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

// Per-image chains (the schemes' elementwise=True mode, image_augmentations.py:563-570 / augmentation_schemes.py:135): items_dev holds
// one FusedOp per (level, image), [n_ops][B].  One final launch for the whole batch (+ a histogram pass and a table launch per level at
// which some image has an AutoContrast / Equalize: `table_levels` bit l), every workgroup evaluating its own image's chain.
// Sorting an elementwise batch by what its chains need (host).  Kind 0: pixel-local chains (fused_local_kernel), 1: no Sharpness and
// warps that keep a row a row (FUSED_ROWS), 2: the rest (general evaluators; the chains with a Sharpness first - their workgroups run
// longest, the short ones fill the tail).  Row 0 of `order_out` holds all images: kinds 0 .. 2 of the chains WITHOUT a table op, then
// kinds 0 .. 2 of the chains with one (groups 3 .. 5: their final launch waits for the histogram passes, the others start at once);
// row 1 + l the images with a table op at level l by the kind of the levels UNDER it (what the histogram pass evaluates; groups 0 .. 2).
int chb_aug_items_sort(void* recs_host, int B, int H, int W, int n_ops, int32_t* order_out, int32_t* counts_out) {
    if (B == 0) return CHB_OK;
    if (!recs_host || !order_out || !counts_out || B < 0 || H <= 0 || W <= 0 || n_ops < 1 || n_ops > CHB_FUSED_MAX_OPS) return CHB_EINVAL;
    FusedOp* recs = (FusedOp*)recs_host;
    int budget = 32;            // pure-shift checks this call pays for (the schemes draw signs, not magnitudes: a handful of records)
    for (int l = 0; l < n_ops; ++l)
        for (int n = 0; n < B; ++n) {
            FusedOp& o = recs[(int64_t)l * B + n];
            if (o.op < CHB_AUG_IDENTITY || o.op > CHB_AUG_CUTOUT) return CHB_EINVAL;
            o.pad = (o.op == CHB_AUG_AFFINE && o.f[3] == 0.0f && affine_row_contiguous(o, H, W, &budget)) ? 1 : 0;
        }
    std::vector<int> key(B);       // 2 * group + (0: a Sharpness in the evaluated levels, 1: none), -1: not in this row
    for (int row = 0; row <= n_ops; ++row) {
        const int depth = row == 0 ? n_ops : row - 1;          // the levels the launch evaluates
        int cnt[2 * CHB_ITEMS_GROUPS];
        for (int k = 0; k < 2 * CHB_ITEMS_GROUPS; ++k) cnt[k] = 0;
        for (int n = 0; n < B; ++n) {
            key[n] = -1;
            if (row > 0) {
                const int op = recs[(int64_t)(row - 1) * B + n].op;
                if (op != CHB_AUG_AUTOCONTRAST && op != CHB_AUG_EQUALIZE) continue;
            }
            bool local = true, rows = true, sharp = false, table = false;
            for (int l = 0; l < depth; ++l) {
                const FusedOp& o = recs[(int64_t)l * B + n];
                local = local && fused_is_local(o.op);
                rows = rows && o.op != CHB_AUG_SHARPNESS && (o.op != CHB_AUG_AFFINE || o.f[3] == 0.0f);
                sharp = sharp || o.op == CHB_AUG_SHARPNESS;
                table = table || o.op == CHB_AUG_AUTOCONTRAST || o.op == CHB_AUG_EQUALIZE;
            }
            const int g = (local ? 0 : (rows ? 1 : 2)) + ((row == 0 && table) ? 3 : 0);
            key[n] = 2 * g + (sharp ? 0 : 1);
            ++cnt[key[n]];
        }
        int at[2 * CHB_ITEMS_GROUPS], run = 0;
        for (int k = 0; k < 2 * CHB_ITEMS_GROUPS; ++k) { at[k] = run; run += cnt[k]; }
        int32_t* ord = order_out + (int64_t)row * B;
        for (int n = 0; n < B; ++n) ord[n] = 0;
        for (int n = 0; n < B; ++n)
            if (key[n] >= 0) ord[at[key[n]]++] = n;
        for (int g = 0; g < CHB_ITEMS_GROUPS; ++g) counts_out[row * CHB_ITEMS_GROUPS + g] = cnt[2 * g] + cnt[2 * g + 1];
    }
    return CHB_OK;
}

// The launches of a sorted batch are independent group by group, and each fills a fraction of the chip: the chains without a table op
// start at once on two side streams - the general ones, the long pole, alone on theirs - while the caller's stream and a third side
// stream walk the table levels (histogram passes, then the table launch) and then run the chains with tables; forked and joined with
// events (inside a stream capture these become parallel branches of the graph).  Three side streams and no more: the runtime has four
// hardware queues, a fifth stream shares one and serialises behind whatever runs there (measured: 234 us against 215).  One set per device, made on first use; the enqueue section holds the lock
// (the events are shared).
struct GroupStreams {
    hipStream_t side[3];
    hipEvent_t fork, join[3];
    bool ok;
};
static std::mutex g_group_mu;
static GroupStreams* group_streams() {
    static GroupStreams sets[64];
    static bool made[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!made[dev]) {
        GroupStreams& g = sets[dev];
        g.ok = hipEventCreateWithFlags(&g.fork, hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k < 3; ++k)
            g.ok = g.ok && hipStreamCreateWithFlags(&g.side[k], hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&g.join[k], hipEventDisableTiming) == hipSuccess;
        made[dev] = true;
    }
    return sets[dev].ok ? &sets[dev] : nullptr;
}

// order_dev / counts: NULL (every image through the general evaluators, one launch), or what chb_aug_items_sort made of the records
static int fused_items_run(const uint8_t* in, void* out, int B, int H, int W, int n_ops, const void* items_dev, const int32_t* const* centers_dev,
                           int table_levels, int32_t* workspace, int patch, const int32_t* order_dev, const int32_t* counts, void* stream) {
    if (B == 0) return CHB_OK;
    if (!in || !out || !items_dev || B < 0 || H <= 0 || W <= 0 || n_ops < 1 || n_ops > CHB_FUSED_MAX_OPS || patch < 0 || (patch & 3)) return CHB_EINVAL;
    if ((int64_t)H * W * 3 >= 2147483647LL - 4 || B > 65535) return CHB_EUNSUPPORTED;
    if (patch && (H / patch == 0 || W / patch == 0)) return CHB_EINVAL;
    if ((table_levels & ((1 << n_ops) - 1)) && !workspace) return CHB_EINVAL;
    if ((order_dev == nullptr) != (counts == nullptr)) return CHB_EINVAL;
    if (counts) {
        for (int row = 0; row <= n_ops; ++row) {
            int tot = 0;
            for (int g = 0; g < CHB_ITEMS_GROUPS; ++g) {
                const int c = counts[row * CHB_ITEMS_GROUPS + g];
                if (c < 0 || (row > 0 && g >= 3 && c)) return CHB_EINVAL;
                tot += c;
            }
            if (row == 0 ? tot != B : tot > B) return CHB_EINVAL;
            if (row > 0 && tot > 0 && !(table_levels & (1 << (row - 1)))) return CHB_EINVAL;
        }
    }
    hipStream_t s0 = (hipStream_t)stream;
    hipStream_t s = s0;
    std::unique_lock<std::mutex> lock(g_group_mu, std::defer_lock);
    GroupStreams* gs = nullptr;
    if (order_dev) {
        lock.lock();
        gs = group_streams();
        if (!gs) return CHB_ELAUNCH;
    }
    auto fork = [&](int first, int last) -> bool {            // side streams first .. last wait for what the caller's stream holds so far
        if (hipEventRecord(gs->fork, s0) != hipSuccess) return false;
        for (int k = first; k <= last; ++k)
            if (hipStreamWaitEvent(gs->side[k], gs->fork, 0) != hipSuccess) return false;
        return true;
    };
    auto join = [&](int first, int last) -> bool {            // ... and the caller's stream for them
        for (int k = first; k <= last; ++k)
            if (hipEventRecord(gs->join[k], gs->side[k]) != hipSuccess || hipStreamWaitEvent(s0, gs->join[k], 0) != hipSuccess) return false;
        return true;
    };
    // once side streams have been forked, EVERY exit joins them again (best effort on an error path): an unjoined branch invalidates a
    // stream capture, and outside one the caller may free the workspace while a side-stream kernel still uses it
    bool forked = false;
    auto fail = [&]() -> int {
        if (forked) (void)join(0, 2);
        return CHB_ELAUNCH;
    };
    const FusedOp* items = (const FusedOp*)items_dev;
    FusedParams P;
    memset(&P, 0, sizeof(P));
    P.n = n_ops; P.B = B; P.H = H; P.W = W;
    P.items = items;
    const int fast = ((W & 3) == 0 && !((uintptr_t)in & 3) && (patch || !((uintptr_t)out & 3))) ? 1 : 0;
    int n_tables = 0;
    for (int l = 0; l < n_ops; ++l) {
        P.centers[l] = centers_dev ? centers_dev[l] : nullptr;          // the [B,2] cutout centres of the level (images without a CutOut there ignore them)
        if (table_levels & (1 << l)) P.lut[l] = workspace + (int64_t)(n_tables++) * fused_table_ints(B, H, W);
    }
    int gh = 0, gw = 0;
    if (patch) { gh = H / patch; gw = W / patch; }
    const int hh = patch ? gh * patch : H;
#define CHB_ITEMS_FINAL_N(KERNEL_, ...)                                                                                                                \
    switch (n_ops) {                                                                                                                                   \
        case 1: KERNEL_(1, __VA_ARGS__); break;                                                                                                        \
        case 2: KERNEL_(2, __VA_ARGS__); break;                                                                                                        \
        case 3: KERNEL_(3, __VA_ARGS__); break;                                                                                                        \
        default: KERNEL_(4, __VA_ARGS__); break;                                                                                                       \
    }
#define CHB_ITEMS_FINAL(NL, MODE_, GRID_)                                                                                                              \
    do {                                                                                                                                               \
        if (patch) hipLaunchKernelGGL((fused_final_kernel<NL, true, MODE_>), GRID_, dim3(256), 0, s, in, out, Q, patch, gh, gw, fast);                \
        else hipLaunchKernelGGL((fused_final_kernel<NL, false, MODE_>), GRID_, dim3(256), 0, s, in, out, Q, patch, gh, gw, fast);                     \
    } while (0)
#define CHB_ITEMS_LOCAL(NL, GRID_)                                                                                                                     \
    do {                                                                                                                                               \
        if (patch) hipLaunchKernelGGL((fused_local_kernel<NL, true, true>), GRID_, dim3(256), 0, s, in, out, Q, patch, gh, gw, fast);                 \
        else hipLaunchKernelGGL((fused_local_kernel<NL, false, true>), GRID_, dim3(256), 0, s, in, out, Q, patch, gh, gw, fast);                      \
    } while (0)
    // final launch of group g (kind g % 3) of row 0 on stream `on`
    auto final_group = [&](int g, hipStream_t on) {
        const int c = counts[g];
        if (!c) return;
        FusedParams Q = P;
        Q.order = order_dev;
        for (int k = 0; k < g; ++k) Q.n0 += counts[k];
        s = on;
        if (g % 3 == 0) {
            const dim3 grid((hh + 31) / 32, c);
            CHB_ITEMS_FINAL_N(CHB_ITEMS_LOCAL, grid);
        } else if (g % 3 == 1) {
            const dim3 grid((hh + 31) / 32, c);                 // FUSED_ROWS: 8 rows per wave
            CHB_ITEMS_FINAL_N(CHB_ITEMS_FINAL, FUSED_ITEMS_ROWS, grid);
        } else {
            const dim3 grid((hh + 15) / 16, c);
            CHB_ITEMS_FINAL_N(CHB_ITEMS_FINAL, FUSED_ITEMS, grid);
        }
        s = s0;
    };
    if (order_dev) {            // the chains without a table op have nothing to wait for: beside the table levels, the long pole alone
        forked = true;          // a partial fork may already have made a side stream wait
        if (!fork(0, 2)) return fail();
        final_group(2, gs->side[0]);
        final_group(0, gs->side[1]);
        final_group(1, gs->side[1]);
    }
    const int slices = slices_for((int64_t)H * W * 3, B);
    for (int l = 0; l < n_ops; ++l) {
        if (!P.lut[l]) continue;
        int32_t* t = const_cast<int32_t*>(P.lut[l]);
        int32_t* part = t + (int64_t)B * 768;
#define CHB_ITEMS_HIST(NL, MODE_) hipLaunchKernelGGL((fused_hist_kernel<NL, MODE_>), grid, dim3(256), 0, s, in, part, Q, fast, 0, 0, -2, -2, -2)
#define CHB_ITEMS_HIST_L(MODE_)                                                                                                                       \
    switch (l) {                                                                                                                                      \
        case 0: CHB_ITEMS_HIST(0, MODE_); break;                                                                                                      \
        case 1: CHB_ITEMS_HIST(1, MODE_); break;                                                                                                      \
        case 2: CHB_ITEMS_HIST(2, MODE_); break;                                                                                                      \
        default: CHB_ITEMS_HIST(3, MODE_); break;                                                                                                     \
    }
        if (!order_dev) {
            const FusedParams& Q = P;
            const dim3 grid(slices, B);
            CHB_ITEMS_HIST_L(FUSED_ITEMS);
            hipLaunchKernelGGL(fused_lut_kernel, dim3(B * 3), dim3(256), 0, s, t, part, slices, 0, items + (int64_t)l * B, (const int32_t*)nullptr);
            continue;
        }
        // the images with a table op at this level only, in their own launch per kind; fewer images take more slices each - up to 16:
        // the table launch adds them up one after the other (the partial tables of `tot` images x `sl` slices fit where B x slices
        // were reserved)
        const int32_t* cn = counts + (l + 1) * CHB_ITEMS_GROUPS;
        const int tot = cn[0] + cn[1] + cn[2];
        if (tot == 0) continue;
        int sl = slices_for((int64_t)H * W * 3, tot);
        if (sl > 16) sl = 16 > slices ? 16 : slices;
        if ((int64_t)sl * tot > (int64_t)slices * B) sl = (int)(((int64_t)slices * B) / tot);
        FusedParams Q = P;
        Q.order = order_dev + (int64_t)(l + 1) * B;
        const bool spread = (cn[0] != 0) + (cn[1] != 0) + (cn[2] != 0) > 1;
        if (spread && !fork(2, 2)) return fail();
        for (int g = 0; g < 3; ++g) {
            if (cn[g]) {
                const dim3 grid(sl, cn[g]);
                s = (g == 2 || !spread) ? s0 : gs->side[2];
                if (g == 0) { CHB_ITEMS_HIST_L(FUSED_ITEMS_LOCAL); }
                else if (g == 1) { CHB_ITEMS_HIST_L(FUSED_ITEMS_ROWS); }
                else { CHB_ITEMS_HIST_L(FUSED_ITEMS); }
            }
            Q.n0 += cn[g];
        }
        s = s0;
        if (spread && !join(2, 2)) return fail();
        hipLaunchKernelGGL(fused_lut_kernel, dim3(tot * 3), dim3(256), 0, s, t, part, sl, 0, items + (int64_t)l * B, Q.order);
#undef CHB_ITEMS_HIST_L
#undef CHB_ITEMS_HIST
    }
    if (!order_dev) {
        const FusedParams& Q = P;
        const dim3 grid((hh + 15) / 16, B);
        CHB_ITEMS_FINAL_N(CHB_ITEMS_FINAL, FUSED_ITEMS, grid);
    } else {                    // the chains with tables behind the table launches
        if ((counts[3] || counts[4]) && !fork(2, 2)) return fail();
        final_group(5, s0);
        final_group(3, gs->side[2]);
        final_group(4, gs->side[2]);
        if (!join(0, 2)) return fail();
    }
#undef CHB_ITEMS_LOCAL
#undef CHB_ITEMS_FINAL
#undef CHB_ITEMS_FINAL_N
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_aug_fused_items(const uint8_t* in, void* out, int B, int H, int W, int n_ops, const void* items_dev, const int32_t* const* centers_dev,
                        int table_levels, int32_t* workspace, int patch, void* stream) {
    return fused_items_run(in, out, B, H, W, n_ops, items_dev, centers_dev, table_levels, workspace, patch, nullptr, nullptr, stream);
}

int chb_aug_fused_items_sorted(const uint8_t* in, void* out, int B, int H, int W, int n_ops, const void* items_dev, const int32_t* const* centers_dev,
                               int table_levels, int32_t* workspace, int patch, const int32_t* order_dev, const int32_t* counts_host, void* stream) {
    if (B > 0 && (!order_dev || !counts_host)) return CHB_EINVAL;
    return fused_items_run(in, out, B, H, W, n_ops, items_dev, centers_dev, table_levels, workspace, patch, order_dev, counts_host, stream);
}

int chb_aug_sharpness(const uint8_t* in, uint8_t* out, int B, int H, int W, int C, float factor, void* stream) {
    if (B == 0) return CHB_OK;
    if (!in || !out || B < 0 || H <= 0 || W <= 0 || C <= 0 || C > 4) return CHB_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (C == 3 && (W & 3) == 0 && !((uintptr_t)in & 3) && !((uintptr_t)out & 3)) {
        const int grid2 = row_grid((int64_t)B * ((H + 1) / 2));
        if (factor == 0.0f) hipLaunchKernelGGL(sharpness_rows_kernel<0>, dim3(grid2), dim3(256), 0, s, in, out, B, H, W, factor);
        else if (factor > 0.0f && factor < 1.0f) hipLaunchKernelGGL(sharpness_rows_kernel<1>, dim3(grid2), dim3(256), 0, s, in, out, B, H, W, factor);
        else hipLaunchKernelGGL(sharpness_rows_kernel<2>, dim3(grid2), dim3(256), 0, s, in, out, B, H, W, factor);
        CHB_LAUNCH_CHECK();
        return CHB_OK;
    }
    const int grid = row_grid((int64_t)B * H);
    if (factor == 0.0f) hipLaunchKernelGGL(sharpness_kernel<0>, dim3(grid), dim3(256), 0, s, in, out, B, H, W, C, factor);
    else if (factor > 0.0f && factor < 1.0f) hipLaunchKernelGGL(sharpness_kernel<1>, dim3(grid), dim3(256), 0, s, in, out, B, H, W, C, factor);
    else hipLaunchKernelGGL(sharpness_kernel<2>, dim3(grid), dim3(256), 0, s, in, out, B, H, W, C, factor);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_normalize_u8(const uint8_t* in, float* out, int64_t n_pixels, int channels, int mode, void* stream) {
    if (n_pixels == 0) return CHB_OK;
    if (!in || !out || n_pixels < 0 || channels <= 0) return CHB_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nb = n_pixels * channels;
    const bool aligned = !((uintptr_t)in & 3) && !((uintptr_t)out & 15);
    if (mode == CHB_NORM_TF) {
        if (!aligned) return CHB_EINVAL;
        hipLaunchKernelGGL(normalize_u8x4_kernel<1>, dim3(stream_grid((nb / 4 + 3) / 4)), dim3(256), 0, s, in, out, nb / 4, nb, channels, kTorch);
    } else if (channels != 3) {
        return CHB_EINVAL;  // caffe/torch carry 3-channel constants (:649,:656)
    } else if (mode == CHB_NORM_CAFFE) {
        hipLaunchKernelGGL((normalize_kernel<0, uint8_t>), dim3(stream_grid(n_pixels)), dim3(256), 0, s, in, out, n_pixels, kCaffe);
    } else if (mode == CHB_NORM_TORCH) {
        if (aligned) hipLaunchKernelGGL(normalize_u8x4_kernel<2>, dim3(stream_grid((nb / 4 + 3) / 4)), dim3(256), 0, s, in, out, nb / 4, nb, 3, kTorch);
        else hipLaunchKernelGGL((normalize_kernel<2, uint8_t>), dim3(stream_grid(n_pixels)), dim3(256), 0, s, in, out, n_pixels, kTorch);
    } else {
        return CHB_EINVAL;
    }
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_normalize_f32(const float* in, float* out, int64_t n_pixels, int channels, int mode, void* stream) {
    if (n_pixels == 0) return CHB_OK;
    if (!in || !out || n_pixels < 0 || channels != 3) return CHB_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int grid = stream_grid(n_pixels);
    if (mode == CHB_NORM_TF) hipLaunchKernelGGL((normalize_kernel<1, float>), dim3(grid), dim3(256), 0, s, in, out, n_pixels, kCaffe);
    else if (mode == CHB_NORM_CAFFE) hipLaunchKernelGGL((normalize_kernel<0, float>), dim3(grid), dim3(256), 0, s, in, out, n_pixels, kCaffe);
    else if (mode == CHB_NORM_TORCH) hipLaunchKernelGGL((normalize_kernel<2, float>), dim3(grid), dim3(256), 0, s, in, out, n_pixels, kTorch);
    else return CHB_EINVAL;
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_normalize_patchify_bf16(const uint8_t* in, void* out, int B, int H, int W, int patch, int mode, void* stream) {
    if (B == 0) return CHB_OK;
    if (!in || !out || B < 0 || H <= 0 || W <= 0 || patch <= 0 || (patch & 3)) return CHB_EINVAL;
    const int gh = H / patch, gw = W / patch;
    if (gh == 0 || gw == 0) return CHB_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    bf16_t* o = (bf16_t*)out;
    if (B > 65535) return CHB_EUNSUPPORTED;
    static const int rows_env = getenv("CHB_NP_ROWS") ? atoi(getenv("CHB_NP_ROWS")) : 8;      // rows per wave (A/B: 4, 8, 16)
#define CHB_NP_LAUNCH(MODE_, NC_)                                                                                                                  \
    do {                                                                                                                                           \
        if (rows_env == 4) hipLaunchKernelGGL((normalize_patchify_kernel<MODE_, 4>), dim3((gh * patch + 15) / 16, B), dim3(256), 0, s, in, o, B, H, W, patch, gh, gw, NC_);       \
        else if (rows_env == 16) hipLaunchKernelGGL((normalize_patchify_kernel<MODE_, 16>), dim3((gh * patch + 63) / 64, B), dim3(256), 0, s, in, o, B, H, W, patch, gh, gw, NC_); \
        else hipLaunchKernelGGL((normalize_patchify_kernel<MODE_, 8>), dim3((gh * patch + 31) / 32, B), dim3(256), 0, s, in, o, B, H, W, patch, gh, gw, NC_);                    \
    } while (0)
    if (mode == CHB_NORM_TF) CHB_NP_LAUNCH(1, kCaffe);
    else if (mode == CHB_NORM_CAFFE) CHB_NP_LAUNCH(0, kCaffe);
    else if (mode == CHB_NORM_TORCH) CHB_NP_LAUNCH(2, kTorch);
    else return CHB_EINVAL;
#undef CHB_NP_LAUNCH
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_patchify_f32_bf16(const float* in, void* out, int B, int H, int W, int patch, void* stream) {
    if (B == 0) return CHB_OK;
    if (!in || !out || B < 0 || H <= 0 || W <= 0 || patch <= 0 || (patch & 3)) return CHB_EINVAL;
    const int gh = H / patch, gw = W / patch;
    if (gh == 0 || gw == 0) return CHB_EINVAL;
    const int64_t total = (int64_t)B * gh * patch * ((gw * patch) / 4);
    hipLaunchKernelGGL(patchify_f32_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, B, H, W, patch,
                       gh, gw);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

}  // extern "C"
