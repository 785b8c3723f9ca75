// Input side of the augmentation path: the Keras preprocessing layers the reference re-exports as
// chambers.augmentations.{Resizing, CenterCrop, RandomCrop, RandomFlip, Rescaling} (augmentations/__init__.py:1-13) and its
// own ResizingMinMax (augmentations/image_augmentations.py:686-748), on NHWC batches resident in HBM.
//   resize      : tf.image.resize, TF2 semantics [UPSTREAM-RECALLED]: half-pixel centres, no antialias.
//                 bilinear -> float32 out, fp32 arithmetic in the order of the CPU kernel (built with -ffp-contract=off):
//                   in = (out + 0.5) * scale - 0.5; lo = max(floor(in), 0); hi = min(ceil(in), size - 1); t = in - floor(in)
//                   top = tl + (tr - tl) * tx; bot = bl + (br - bl) * tx; out = top + (bot - top) * ty
//                 nearest  -> input dtype, index min(floor((out + 0.5) * scale), size - 1)
//   crop / flip : one gather kernel: per-image window offset (CenterCrop: one for all, RandomCrop: one draw for the batch) and
//                 per-image horizontal / vertical flip bits (RandomFlip flips every image independently).
//   rescale     : float32(x) * scale + offset.
// All are HBM-bound: algorithmic bytes = input window read once + output written once.
#include "common.hpp"
#include "../../include/chambers_hip.h"

namespace {

struct __attribute__((packed)) u32_unaligned { uint32_t v; };

template <typename TIN>
__device__ __forceinline__ float ld(const TIN* p) { return (float)*p; }

// one thread = one output pixel (all C channels); 64-bit flat index over B * OH * OW
template <typename TIN>
__global__ void __launch_bounds__(256) resize_bilinear_kernel(const TIN* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                                              int C, int OH, int OW, float sy, float sx) {
    const int64_t total = (int64_t)B * OH * OW;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % OW);
        const int64_t r = idx / OW;
        const int oy = (int)(r % OH), b = (int)(r / OH);
        const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
        const float fy0 = floorf(fy), fx0 = floorf(fx);
        const int y0 = max((int)fy0, 0), y1 = min((int)ceilf(fy), H - 1);
        const int x0 = max((int)fx0, 0), x1 = min((int)ceilf(fx), W - 1);
        const float ty = fy - fy0, tx = fx - fx0;
        const TIN* img = in + (int64_t)b * H * W * C;
        const TIN* p00 = img + ((int64_t)y0 * W + x0) * C;
        const TIN* p01 = img + ((int64_t)y0 * W + x1) * C;
        const TIN* p10 = img + ((int64_t)y1 * W + x0) * C;
        const TIN* p11 = img + ((int64_t)y1 * W + x1) * C;
        float* o = out + idx * C;
        for (int c = 0; c < C; ++c) {
            const float tl = ld(p00 + c), tr = ld(p01 + c), bl = ld(p10 + c), br = ld(p11 + c);
            const float top = tl + (tr - tl) * tx;
            const float bot = bl + (br - bl) * tx;
            o[c] = top + (bot - top) * ty;
        }
    }
}

// uint8 RGB fast path: one thread = 4 consecutive output pixels of a row (12 floats = three float4 stores); every source pixel is
// one unaligned dword load (3 payload bytes).  Same arithmetic, same order as the generic kernel.
__global__ void __launch_bounds__(256) resize_bilinear_rgb8_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int B, int H,
                                                                   int W, int OH, int OW, float sy, float sx) {
    const int wq = OW >> 2;
    const int64_t total = (int64_t)B * OH * wq;
    const int64_t total_bytes = (int64_t)B * H * W * 3;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int xq = (int)(idx % wq);
        const int64_t r = idx / wq;
        const int oy = (int)(r % OH), b = (int)(r / OH);
        const float fy = ((float)oy + 0.5f) * sy - 0.5f;
        const float fy0 = floorf(fy);
        const int y0 = max((int)fy0, 0), y1 = min((int)ceilf(fy), H - 1);
        const float ty = fy - fy0;
        const int64_t row0 = ((int64_t)b * H + y0) * W * 3, row1 = ((int64_t)b * H + y1) * W * 3;
        float f[12];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ox = 4 * xq + i;
            const float fx = ((float)ox + 0.5f) * sx - 0.5f;
            const float fx0 = floorf(fx);
            const int x0 = max((int)fx0, 0), x1 = min((int)ceilf(fx), W - 1);
            const float tx = fx - fx0;
            uint32_t px[4];
            const int64_t offs[4] = {row0 + (int64_t)x0 * 3, row0 + (int64_t)x1 * 3, row1 + (int64_t)x0 * 3, row1 + (int64_t)x1 * 3};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int over = (offs[k] + 4 > total_bytes) ? 1 : 0;     // last pixel of the batch: step back one byte and shift
                px[k] = reinterpret_cast<const u32_unaligned*>(in + offs[k] - over)->v >> (8 * over);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float tl = (float)((px[0] >> (8 * c)) & 0xffu), tr = (float)((px[1] >> (8 * c)) & 0xffu);
                const float bl = (float)((px[2] >> (8 * c)) & 0xffu), br = (float)((px[3] >> (8 * c)) & 0xffu);
                const float top = tl + (tr - tl) * tx;
                const float bot = bl + (br - bl) * tx;
                f[3 * i + c] = top + (bot - top) * ty;
            }
        }
        float4* o = reinterpret_cast<float4*>(out + (((int64_t)b * OH + oy) * OW + 4 * xq) * 3);
        o[0] = make_float4(f[0], f[1], f[2], f[3]);
        o[1] = make_float4(f[4], f[5], f[6], f[7]);
        o[2] = make_float4(f[8], f[9], f[10], f[11]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) resize_nearest_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C,
                                                             int OH, int OW, float sy, float sx) {
    const int64_t total = (int64_t)B * OH * OW;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % OW);
        const int64_t r = idx / OW;
        const int oy = (int)(r % OH), b = (int)(r / OH);
        const int iy = min((int)floorf(((float)oy + 0.5f) * sy), H - 1);
        const int ix = min((int)floorf(((float)ox + 0.5f) * sx), W - 1);
        const T* p = in + (((int64_t)b * H + iy) * W + ix) * C;
        T* o = out + idx * C;
        for (int c = 0; c < C; ++c) o[c] = p[c];
    }
}

// Ragged batch (the decode -> Resizing step of data/dataset.py + augmentations/__init__.py:11): B decoded RGB
// images of different sizes, packed back to back in one uint8 buffer (`offs[b]` = byte offset of image b, `hw[2b], hw[2b+1]` = its
// height and width), are resized to one [B, OH, OW, 3] batch in a single launch -- per-image scales H/OH, W/OW, otherwise the
// arithmetic of the kernels above.  One thread = 4 consecutive output pixels of a row.  OUT: fp32 (tf.image.resize) or uint8
// (tf.cast(float -> uint8) of it: truncation), the layout the augmentation kernels take.
template <bool BILINEAR, typename TOUT>
__global__ void __launch_bounds__(256) resize_ragged_rgb8_kernel(const uint8_t* __restrict__ in, const int64_t* __restrict__ offs,
                                                                 const int32_t* __restrict__ hw, TOUT* __restrict__ out, int B, int OH,
                                                                 int OW, int64_t total_bytes) {
    const int wq = OW >> 2;
    const int64_t total = (int64_t)B * OH * wq;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int xq = (int)(idx % wq);
        const int64_t r = idx / wq;
        const int oy = (int)(r % OH), b = (int)(r / OH);
        const int H = hw[2 * b], W = hw[2 * b + 1];
        const int64_t base = offs[b];
        const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
        float f[12];
        if (BILINEAR) {
            const float fy = ((float)oy + 0.5f) * sy - 0.5f;
            const float fy0 = floorf(fy);
            const int y0 = max((int)fy0, 0), y1 = min((int)ceilf(fy), H - 1);
            const float ty = fy - fy0;
            const int64_t row0 = base + (int64_t)y0 * W * 3, row1 = base + (int64_t)y1 * W * 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ox = 4 * xq + i;
                const float fx = ((float)ox + 0.5f) * sx - 0.5f;
                const float fx0 = floorf(fx);
                const int x0 = max((int)fx0, 0), x1 = min((int)ceilf(fx), W - 1);
                const float tx = fx - fx0;
                uint32_t px[4];
                const int64_t o4[4] = {row0 + (int64_t)x0 * 3, row0 + (int64_t)x1 * 3, row1 + (int64_t)x0 * 3, row1 + (int64_t)x1 * 3};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int over = (o4[k] + 4 > total_bytes) ? 1 : 0;     // last pixel of the buffer: step back one byte and shift
                    px[k] = reinterpret_cast<const u32_unaligned*>(in + o4[k] - over)->v >> (8 * over);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float tl = (float)((px[0] >> (8 * c)) & 0xffu), tr = (float)((px[1] >> (8 * c)) & 0xffu);
                    const float bl = (float)((px[2] >> (8 * c)) & 0xffu), br = (float)((px[3] >> (8 * c)) & 0xffu);
                    const float top = tl + (tr - tl) * tx;
                    const float bot = bl + (br - bl) * tx;
                    f[3 * i + c] = top + (bot - top) * ty;
                }
            }
        } else {
            const int iy = min((int)floorf(((float)oy + 0.5f) * sy), H - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ix = min((int)floorf(((float)(4 * xq + i) + 0.5f) * sx), W - 1);
                const int64_t o1 = base + ((int64_t)iy * W + ix) * 3;
                const int over = (o1 + 4 > total_bytes) ? 1 : 0;
                const uint32_t px = reinterpret_cast<const u32_unaligned*>(in + o1 - over)->v >> (8 * over);
#pragma unroll
                for (int c = 0; c < 3; ++c) f[3 * i + c] = (float)((px >> (8 * c)) & 0xffu);
            }
        }
        const int64_t o0 = (((int64_t)b * OH + oy) * OW + 4 * xq) * 3;
        if (sizeof(TOUT) == 4) {
            float4* o = reinterpret_cast<float4*>(out + o0);
            o[0] = make_float4(f[0], f[1], f[2], f[3]);
            o[1] = make_float4(f[4], f[5], f[6], f[7]);
            o[2] = make_float4(f[8], f[9], f[10], f[11]);
        } else {
            uint32_t w[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                w[k] = ((uint32_t)(int)f[4 * k] & 0xffu) | (((uint32_t)(int)f[4 * k + 1] & 0xffu) << 8) |
                       (((uint32_t)(int)f[4 * k + 2] & 0xffu) << 16) | (((uint32_t)(int)f[4 * k + 3] & 0xffu) << 24);
            uint32_t* o = reinterpret_cast<uint32_t*>(out + o0);
            o[0] = w[0]; o[1] = w[1]; o[2] = w[2];
        }
    }
}

// crop + flips as a gather of whole pixels of `psz` bytes.  One wave per output row (no per-thread division); uint8 RGB rows
// move 4 pixels (12 bytes) per lane with one unaligned dword load per source pixel, everything else byte-wise.
__global__ void __launch_bounds__(256) crop_flip_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int H, int W,
                                                        int psz, int OH, int OW, const int32_t* __restrict__ offsets, int per_image,
                                                        int oy0, int ox0, const uint8_t* __restrict__ flips) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int rows = B * OH;
    const int64_t total_bytes = (int64_t)B * H * W * psz;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const int b = row / OH, oy = row - b * OH;
        int y0 = oy0, x0 = ox0;
        if (offsets) {
            const int32_t* o = offsets + (per_image ? 2 * b : 0);
            y0 = o[0]; x0 = o[1];
        }
        const int fl = flips ? flips[b] : 0;
        const int sy = y0 + ((fl & 2) ? OH - 1 - oy : oy);
        const int64_t src_row = (((int64_t)b * H + sy) * W) * psz;
        uint8_t* orow = out + (int64_t)row * OW * psz;
        if (psz == 3 && (OW & 3) == 0 && !((uintptr_t)out & 3)) {
            for (int xq = lane; xq < (OW >> 2); xq += 64) {
                uint32_t w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ox = 4 * xq + i;
                    const int sx = x0 + ((fl & 1) ? OW - 1 - ox : ox);
                    const int64_t off = src_row + (int64_t)sx * 3;
                    const int over = (off + 4 > total_bytes) ? 1 : 0;      // last pixel of the batch: step back one byte and shift
                    w[i] = reinterpret_cast<const u32_unaligned*>(in + off - over)->v >> (8 * over);
                }
                uint32_t o0 = (w[0] & 0xffffffu) | (w[1] << 24);
                uint32_t o1 = ((w[1] >> 8) & 0xffffu) | (w[2] << 16);
                uint32_t o2 = ((w[2] >> 16) & 0xffu) | (w[3] << 8);
                uint32_t* dst = reinterpret_cast<uint32_t*>(orow + (int64_t)xq * 12);
                dst[0] = o0; dst[1] = o1; dst[2] = o2;
            }
        } else {
            for (int ox = lane; ox < OW; ox += 64) {
                const int sx = x0 + ((fl & 1) ? OW - 1 - ox : ox);
                const uint8_t* s = in + src_row + (int64_t)sx * psz;
                uint8_t* d = orow + (int64_t)ox * psz;
                for (int k = 0; k < psz; ++k) d[k] = s[k];
            }
        }
    }
}

template <typename TIN>
__global__ void __launch_bounds__(256) rescale_kernel(const TIN* __restrict__ in, float* __restrict__ out, int64_t n, float scale, float offset) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = (float)in[i] * scale;
        out[i] = v + offset;
    }
}

// uint8: a lane converts one aligned dword into one float4 (256-byte loads, 1 KiB stores per wave)
__global__ void __launch_bounds__(256) rescale_u8x4_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t n_quads, int64_t n,
                                                           float scale, float offset) {
    const uint32_t* in4 = reinterpret_cast<const uint32_t*>(in);
    float4* out4 = reinterpret_cast<float4*>(out);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t w = in4[q];
        float f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = (float)((w >> (8 * k)) & 0xffu) * scale;
            f[k] = v + offset;
        }
        out4[q] = make_float4(f[0], f[1], f[2], f[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = n_quads * 4; i < n; ++i) {
            const float v = (float)in[i] * scale;
            out[i] = v + offset;
        }
}

inline int grid_1d(int64_t n, int cap = 16384) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" {

int chb_resize(const void* in, int in_dtype, void* out, int B, int H, int W, int C, int OH, int OW, int method, void* stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return CHB_EINVAL;
    if (in_dtype != CHB_DT_U8 && in_dtype != CHB_DT_F32) return CHB_EINVAL;
    if (method != CHB_RESIZE_BILINEAR && method != CHB_RESIZE_NEAREST) return CHB_EUNSUPPORTED;
    if (B == 0) return CHB_OK;
    if (!in || !out) return CHB_EINVAL;
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    const dim3 grid(grid_1d((int64_t)B * OH * OW)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (method == CHB_RESIZE_BILINEAR && in_dtype == CHB_DT_U8 && C == 3 && (OW & 3) == 0 && !((uintptr_t)out & 15)) {
        hipLaunchKernelGGL(resize_bilinear_rgb8_kernel, dim3(grid_1d((int64_t)B * OH * (OW / 4))), block, 0, s, (const uint8_t*)in, (float*)out,
                           B, H, W, OH, OW, sy, sx);
    } else if (method == CHB_RESIZE_BILINEAR) {
        if (in_dtype == CHB_DT_U8) hipLaunchKernelGGL(resize_bilinear_kernel<uint8_t>, grid, block, 0, s, (const uint8_t*)in, (float*)out, B, H, W, C, OH, OW, sy, sx);
        else hipLaunchKernelGGL(resize_bilinear_kernel<float>, grid, block, 0, s, (const float*)in, (float*)out, B, H, W, C, OH, OW, sy, sx);
    } else {
        if (in_dtype == CHB_DT_U8) hipLaunchKernelGGL(resize_nearest_kernel<uint8_t>, grid, block, 0, s, (const uint8_t*)in, (uint8_t*)out, B, H, W, C, OH, OW, sy, sx);
        else hipLaunchKernelGGL(resize_nearest_kernel<float>, grid, block, 0, s, (const float*)in, (float*)out, B, H, W, C, OH, OW, sy, sx);
    }
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_resize_ragged(const void* packed_u8, int64_t packed_bytes, const int64_t* offsets_dev, const int32_t* hw_dev, int B, void* out,
                      int out_dtype, int OH, int OW, int method, void* stream) {
    if (B < 0 || OH <= 0 || OW <= 0 || packed_bytes < 0) return CHB_EINVAL;
    if (out_dtype != CHB_DT_U8 && out_dtype != CHB_DT_F32) return CHB_EINVAL;
    if (method != CHB_RESIZE_BILINEAR && method != CHB_RESIZE_NEAREST) return CHB_EUNSUPPORTED;
    if (OW & 3) return CHB_EUNSUPPORTED;          // a lane writes 4 pixels (12 bytes / 48 bytes, aligned)
    if (B == 0) return CHB_OK;
    if (!packed_u8 || !offsets_dev || !hw_dev || !out || packed_bytes < 4 || ((uintptr_t)out & 15)) return CHB_EINVAL;
    const dim3 grid(grid_1d((int64_t)B * OH * (OW / 4))), block(256);
    hipStream_t s = (hipStream_t)stream;
    const uint8_t* in = (const uint8_t*)packed_u8;
    if (method == CHB_RESIZE_BILINEAR) {
        if (out_dtype == CHB_DT_F32) hipLaunchKernelGGL((resize_ragged_rgb8_kernel<true, float>), grid, block, 0, s, in, offsets_dev, hw_dev, (float*)out, B, OH, OW, packed_bytes);
        else hipLaunchKernelGGL((resize_ragged_rgb8_kernel<true, uint8_t>), grid, block, 0, s, in, offsets_dev, hw_dev, (uint8_t*)out, B, OH, OW, packed_bytes);
    } else {
        if (out_dtype == CHB_DT_F32) hipLaunchKernelGGL((resize_ragged_rgb8_kernel<false, float>), grid, block, 0, s, in, offsets_dev, hw_dev, (float*)out, B, OH, OW, packed_bytes);
        else hipLaunchKernelGGL((resize_ragged_rgb8_kernel<false, uint8_t>), grid, block, 0, s, in, offsets_dev, hw_dev, (uint8_t*)out, B, OH, OW, packed_bytes);
    }
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_crop_flip(const void* in, void* out, int B, int H, int W, int pixel_bytes, int OH, int OW, const int32_t* offsets_dev,
                  int per_image, int oy0, int ox0, const uint8_t* flips_dev, void* stream) {
    if (B < 0 || H <= 0 || W <= 0 || pixel_bytes <= 0 || OH <= 0 || OW <= 0 || OH > H || OW > W) return CHB_EINVAL;
    if (!offsets_dev && (oy0 < 0 || ox0 < 0 || oy0 + OH > H || ox0 + OW > W)) return CHB_EINVAL;
    if (B == 0) return CHB_OK;
    if (!in || !out) return CHB_EINVAL;
    int64_t blocks = ((int64_t)B * OH + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(crop_flip_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)in, (uint8_t*)out, B, H, W,
                       pixel_bytes, OH, OW, offsets_dev, per_image, oy0, ox0, flips_dev);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_rescale(const void* in, int in_dtype, float* out, int64_t n, float scale, float offset, void* stream) {
    if (n < 0 || (in_dtype != CHB_DT_U8 && in_dtype != CHB_DT_F32)) return CHB_EINVAL;
    if (n == 0) return CHB_OK;
    if (!in || !out) return CHB_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == CHB_DT_U8 && !((uintptr_t)in & 3) && !((uintptr_t)out & 15))
        hipLaunchKernelGGL(rescale_u8x4_kernel, dim3(grid_1d(n / 4)), dim3(256), 0, s, (const uint8_t*)in, out, n / 4, n, scale, offset);
    else if (in_dtype == CHB_DT_U8)
        hipLaunchKernelGGL(rescale_kernel<uint8_t>, dim3(grid_1d(n)), dim3(256), 0, s, (const uint8_t*)in, out, n, scale, offset);
    else
        hipLaunchKernelGGL(rescale_kernel<float>, dim3(grid_1d(n)), dim3(256), 0, s, (const float*)in, out, n, scale, offset);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

}  // extern "C"
