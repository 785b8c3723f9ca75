/* chambers_hip.h — C ABI of libchambers_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the ONE hot path of chjort/chambers this build accelerates:
 * RandAugment/AutoAugment -> ImageNetNormalization -> VisionTransformer forward/backward
 * -> AdamW.  The reference has no FFI of its own (it is pure Python over TensorFlow); the
 * interface each entry replaces is therefore the TF/Keras/TFA op sequence issued by the
 * cited reference lines (paths relative to /root/reference/chambers/).  INTEGRATION.md
 * shows the ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - the caller owns all memory (inputs, outputs, workspaces); the library never
 *     allocates, frees or synchronises;
 *   - every entry is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant
 *     and thread-safe for distinct streams; randomness is passed in (decision values or
 *     a 32-bit dropout site key), never drawn from hidden state;
 *   - return value: 0 = ok, CHB_EINVAL (-1) bad argument, CHB_ELAUNCH (-2) launch failure,
 *     CHB_EUNSUPPORTED (-3) shape outside what the kernels are built for.  Nothing throws.
 *   - bf16 tensors are raw uint16 bit patterns (void*); images are uint8 NHWC; token
 *     matrices are row-major [rows, ld].
 */
#ifndef CHAMBERS_HIP_H
#define CHAMBERS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pointwise op ids for chb_aug_pointwise */
#define CHB_PW_INVERT 0       /* augmentations/image_augmentations.py:112-113 */
#define CHB_PW_POSTERIZE 1    /* :171-174   i0 = shift (8 - bits, clamped to 7) */
#define CHB_PW_SOLARIZE 2     /* :192-193   i0 = threshold (may be 256) */
#define CHB_PW_SOLARIZE_ADD 3 /* :212-215   i0 = threshold, i1 = addition */
#define CHB_PW_BRIGHTNESS 4   /* :283-285   factor */
#define CHB_PW_CONTRAST 5     /* :253-265   factor, i0 = degenerate constant (host: pixels/256 clipped) */
#define CHB_PW_COLOR 6        /* :233-235   factor; 3 channels */

/* per-image op ids for chb_aug_dispatch (one 64-byte record per image, see chb_aug_dispatch) */
#define CHB_AUG_IDENTITY 0      /* RandomChance not taken (:522-529) */
#define CHB_AUG_AUTOCONTRAST 1  /* :63-90 */
#define CHB_AUG_EQUALIZE 2      /* :94-103 */
#define CHB_AUG_INVERT 3        /* :107-116 */
#define CHB_AUG_POSTERIZE 4     /* i0 = shift */
#define CHB_AUG_SOLARIZE 5      /* i0 = threshold */
#define CHB_AUG_SOLARIZE_ADD 6  /* i0 = threshold, i1 = addition */
#define CHB_AUG_BRIGHTNESS 7    /* f[0] = factor */
#define CHB_AUG_CONTRAST 8      /* f[0] = factor, i0 = degenerate constant of a batch-1 tensor (H*W/256 clipped, :260-262) */
#define CHB_AUG_COLOR 9         /* f[0] = factor */
#define CHB_AUG_SHARPNESS 10    /* f[0] = factor */
#define CHB_AUG_AFFINE 11       /* f[0..5] = a0 a1 a2 b0 b1 b2 (output -> input, tfa.image.transform), i0 = fill value */
#define CHB_AUG_CUTOUT 12       /* i0 = cy, i1 = cx, i2 = mask_size / 2, i3 = constant value */

#define CHB_NORM_CAFFE 0 /* :647-650 */
#define CHB_NORM_TF 1    /* :659-665 */
#define CHB_NORM_TORCH 2 /* :652-657 */

/* GEMM epilogue modes (chb_gemm_nt) */
#define CHB_EPI_NONE 0   /* C = acc (+bias) */
#define CHB_EPI_GELU 1   /* x = acc+bias; C = gelu(x); aux = bf16(gelu'(x))   layers/transformer.py:42-44 */
#define CHB_EPI_DGELU 2  /* C = acc * aux                                     backward of the above */
#define CHB_EPI_RESID 3  /* C = resid + dropout(acc+bias)                 layers/transformer.py:57-58,69,76 */
#define CHB_EPI_PATCH 4  /* C[row'] = dropout(acc+bias+pos[1+p])          vision_transformer.py:235-261 */

#define CHB_OUT_BF16 0
#define CHB_OUT_F32 1

/* ---------------------------------------------------------------- library info */
int chb_version(void);            /* ABI version, currently 4 (round 4: block-level entries, launch profiler) */
const char* chb_build_arch(void); /* "gfx950" */

/* ---------------------------------------------------------------- augmentation (uint8 NHWC) */
/* Invert / Posterize / Solarize / SolarizeAdd / Brightness / Contrast / Color over a flat
 * NHWC byte buffer (image_augmentations.py:107-293 + blend :10-49).  in == out allowed. */
int chb_aug_pointwise(const uint8_t* in, uint8_t* out, int64_t n_bytes, int op, float factor, int i0, int i1,
                      void* stream);

/* tfa.image.transform / translate / rotate with interpolation="nearest", fill_mode="constant"
 * (ShearX/ShearY/TranslateX/TranslateY/Rotate, image_augmentations.py:120-160,316-484).
 * One 8-float projective transform maps OUTPUT (x,y) to INPUT coordinates.  Pass either
 * transform_host8 (8 floats on the host, shared by the batch, copied into the launch) or
 * transforms_dev ([B,8] if per_image else [1,8], device). */
int chb_aug_affine(const uint8_t* in, uint8_t* out, int B, int H, int W, int C, const float* transform_host8,
                   const float* transforms_dev, int per_image, int fill, void* stream);

/* tfa.image.random_cutout with the per-image centres drawn by the caller: centers_dev is
 * int32 [B,2] = (cy, cx) (CutOut, image_augmentations.py:488-507). mask_size must be even. */
int chb_aug_cutout(const uint8_t* in, uint8_t* out, int B, int H, int W, int C, const int32_t* centers_dev, int mask_size,
                   int value, void* stream);

/* AutoContrast (image_augmentations.py:63-90). workspace: int32 [B*C*2]. */
int chb_aug_autocontrast(const uint8_t* in, uint8_t* out, int B, int H, int W, int C, int32_t* workspace, void* stream);

/* Equalize -> tfa.image.equalize (image_augmentations.py:94-103). workspace: int32 [B*C*256]. */
int chb_aug_equalize(const uint8_t* in, uint8_t* out, int B, int H, int W, int C, int32_t* workspace, void* stream);

/* Sharpness -> tfa.image.sharpness (image_augmentations.py:297-312). in != out. */
int chb_aug_sharpness(const uint8_t* in, uint8_t* out, int B, int H, int W, int C, float factor, void* stream);

/* ImageNetNormalization (image_augmentations.py:621-682), fp32 NHWC output. */
int chb_normalize_u8(const uint8_t* in, float* out, int64_t n_pixels, int channels, int mode, void* stream);
int chb_normalize_f32(const float* in, float* out, int64_t n_pixels, int channels, int mode, void* stream);

/* ImageNetNormalization fused with the patch gather of the patch-embedding Conv2D
 * (vision_transformer.py:235-248,655): uint8 [B,H,W,3] -> bf16 [B*(H/p)*(W/p), p*p*3]. */
int chb_normalize_patchify_bf16(const uint8_t* in, void* out_bf16, int B, int H, int W, int patch, int mode, void* stream);
/* Same gather for an already-normalised float32 NHWC batch (the reference model's own input dtype). */
int chb_patchify_f32_bf16(const float* in, void* out_bf16, int B, int H, int W, int patch, void* stream);

/* ---------------------------------------------------------------- dropout mask (test / tooling) */
/* out[e] = 1 if element e is kept under (key, rate) else 0 — the mask every fused dropout
 * site uses (definition: oracle/rng_ref.py; sites: keras Dropout, layers/transformer.py:38,48). */
int chb_dropout_mask(uint8_t* out, int64_t n, float rate, uint32_t key, void* stream);

/* ---------------------------------------------------------------- ViT block */
/* C[M,N] = epilogue(A[M,K] . B[N,K]^T): bf16 operands, both K-contiguous, fp32 accumulate on
 * MFMA (Dense / einsum projections: layers/attention.py:113-125, layers/transformer.py:72-77,
 * vision_transformer.py:235-283).  bias fp32 [N] or NULL.  out_dtype CHB_OUT_BF16|CHB_OUT_F32.
 *  GELU : aux (bf16 [M,ld_aux]) receives gelu'(acc+bias) (the erf/exp are shared with the forward value).
 *  DGELU: aux is that saved derivative.
 *  RESID: resid fp32 [M,ld_resid] (may alias C); dropout (rate,key) on acc+bias, element
 *         index row*N+col.
 *  PATCH: rows are (image b, patch p) = row / n, row % n with n = period & 0xffffff patches per image and
 *         s = 1 + ((period >> 24) & 15) special tokens in front of them (1: class token, vision_transformer.py:249-256;
 *         2: class + distillation token, :340-357); written to row b*(n+s)+s+p of C; resid = positional table fp32
 *         [n+s, ld_resid]; dropout element index out_row*N+col.
 * out_colsum (fp32 [N], optional): += column sums of C — the bias gradient of the layer that consumes C in the
 * backward chain (fused into the epilogue; caller zeroes it once per step).
 * K % 64 == 0; M, N arbitrary (edges masked); A/B/C 16-byte aligned rows. */
int chb_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                const float* bias, int epilogue, int out_dtype, void* aux, int64_t ld_aux, const float* resid,
                int64_t ld_resid, int period, float drop_rate, uint32_t drop_key, float* out_colsum, void* stream);

/* dW[Kd,Nd] += X[M,Kd]^T . dY[M,Nd]: weight gradient, bf16 operands, fp32 atomic accumulate
 * into dW (caller zeroes it once per step).  M % 64 == 0 (pad rows must be zero). */
int chb_gemm_tn(const void* X, int64_t ldx, const void* dY, int64_t ldy, float* dW, int64_t ldw, int M, int Kd, int Nd,
                void* stream);
/* The same with caller-lent scratch: when `workspace_bytes` >= splits * Kd * Nd * 4 (splits <= number of CUs / number of 256x256
 * output tiles; Kd * Nd * 4 * 256 / tiles bytes always suffice) every split-K work item stores its partial tile with plain
 * 16-byte stores and a second small launch folds the planes into dW, instead of meeting in fp32 atomics (the atomic tail was
 * ~50 us of a 380 us launch).  Falls back to the atomic epilogue when the scratch is too small or misaligned. */
int chb_gemm_tn_ws(const void* X, int64_t ldx, const void* dY, int64_t ldy, float* dW, int64_t ldw, int M, int Kd, int Nd,
                   float* workspace, int64_t workspace_bytes, int fold, float* dy_colsum, void* stream);
/* dy_colsum (fp32 [Nd], optional): += column sums of dY over M — the bias gradient of the same layer — computed inside the GEMM
 * (one more MFMA per dY fragment against an all-ones fragment) instead of a separate pass over dY.
 * fold != 0 above folds right away; with fold == 0 the planes stay in the scratch until this call (same arguments), e.g. to
 * bracket the GEMM launch alone with events.  A no-op when the GEMM of these arguments takes the atomic epilogue. */
int chb_gemm_tn_fold(const float* workspace, int64_t workspace_bytes, float* dW, int64_t ldw, int M, int Kd, int Nd, void* stream);
/* Up to four such folds in ONE launch (a block's four weight gradients, each GEMM run with fold == 0 into a scratch of its own):
 * items_host = HOST array of the arguments chb_gemm_tn_fold would get; per gradient the same sums in the same order. */
typedef struct chb_tn_fold_item {
    const float* workspace;
    int64_t workspace_bytes;
    float* dW;
    int64_t ldw;
    int32_t M, Kd, Nd, reserved;
} chb_tn_fold_item;
int chb_gemm_tn_fold_multi(const chb_tn_fold_item* items_host, int n_items, void* stream);

/* keras LayerNormalization over the last axis (layers/transformer.py:39,49,283): x fp32 rows at
 * stride x_stride, y bf16 [M,D]; mean/rstd fp32 [M] saved for backward. D % 4 == 0, D <= 1024. */
int chb_layernorm_fwd(const float* x, int64_t x_stride, const float* gamma, const float* beta, void* y_bf16,
                      float* mean, float* rstd, int M, int D, float eps, void* stream);
/* dx[row] (+)= LN'(dy); dgamma/dbeta fp32 [D] accumulated with atomics (caller zeroes).
 * Optional fused tail (dz_bf16 != NULL): the backward of the keras Dropout that precedes this residual sum in
 * forward order (layers/transformer.py:69,76) — dz[M,D] = bf16(dx * keep / (1-rate)) (dx_stride must be D), the
 * operand of the next dgrad/wgrad GEMMs, and dz_colsum[D] += its column sums (that layer's bias gradient).
 * zero_gaps != 0 (needs accumulate == 0): the launch also zero-fills the floats between consecutive rows,
 * dx[r*dx_stride + D .. (r+1)*dx_stride) — with pooling="cls" (vision_transformer.py:182-189) only the class rows of the
 * final LayerNorm carry a gradient and every other row of the residual gradient starts at zero. */
int chb_layernorm_bwd(const void* dy_bf16, const float* x, int64_t x_stride, const float* mean, const float* rstd,
                      const float* gamma, float* dx, int64_t dx_stride, int accumulate, float* dgamma, float* dbeta,
                      int M, int D, void* dz_bf16, float* dz_colsum, float drop_rate, uint32_t drop_key, int zero_gaps,
                      void* stream);

/* MultiHeadAttention core (layers/attention.py:7-23,113-125): softmax(QK^T/sqrt(hd)) with
 * dropout on the probabilities, times V.  qkv bf16 [B*N, 3*H*hd] = [Q heads | K heads | V heads];
 * o bf16 [B*N, H*hd]; lse fp32 [B,H,N]. hd == 64; any N with B*H*N*Np4 < 2^32, Np4 = N rounded up to a multiple of 4
 * (the dropout element index ((b*H+h)*N+q)*Np4+k is 32-bit): N <= 224 runs with the head resident in LDS, longer
 * sequences (577 tokens at 384x384) through the streaming forward and the two-pass backward.
 * drop_bits (optional, uint32 [B*H*N*8], N <= 224, drop_rate > 0): the forward writes the keep bits of the mask it applied
 * (word ((bh*N + q)*4 + g)*2 + (t >> 3), bit 16*(r&1) + 8*(r>>1) + (t & 7) for key 16*t + 4*g + r); handed to chb_attention_bwd, the
 * backward tests bits instead of re-hashing (keras Dropout keeps its mask for the backward pass likewise).  NULL: nothing is
 * written / the backward regenerates the mask from the element index.  Same results either way.  With drop_bits and
 * drop_rate > 0 the forward ALWAYS runs the kernel that writes them (the ATTN_FWD_ALGO switch is ignored for that call);
 * N > 224 with drop_bits returns CHB_EUNSUPPORTED (no kernel writes bits there). */
int chb_attention_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int hd, float drop_rate,
                      uint32_t drop_key, uint32_t* drop_bits, void* stream);
/* dbias_qkv (fp32 [3*H*hd], optional): += column sums of dqkv (bias gradient of the fused QKV projection), taken from the
 * fp32 accumulators.  dbias_ws: fp32 [B, 3*H*hd] scratch, required with dbias_qkv when N <= 224 (every head writes its sums to
 * its batch element's row with plain stores and a small second launch folds the rows into dbias_qkv); longer sequences (two-pass
 * kernels) add with one atomic per workgroup and column and ignore it. */
int chb_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int B, int N,
                      int H, int hd, float drop_rate, uint32_t drop_key, float* dbias_qkv, float* dbias_ws,
                      const uint32_t* drop_bits, void* stream);

/* General attention: the rest of the ScaledAttention / MultiHeadAttention call signature, which the ViT never exercises
 * (layers/attention.py:99-153: `mask=[query_mask, value_mask]`, `causal=True`, cross-attention inputs [q, v, k] with Tq != Tv; keras
 * Attention: scores -= 1e9 * (1 - mask), mask = value_mask AND causal lower triangle; dropout on the weights; output *= query_mask).
 * q bf16 [B*Tq, ldq], k / v bf16 [B*Tk, ldk / ldv], head h in columns [h*hd, (h+1)*hd); o bf16 [B*Tq, ldo]; lse fp32 [B, H, Tq];
 * value_mask uint8 [B, Tk], query_mask uint8 [B, Tq] (NULL = all ones).  hd even and <= 128, Tk <= 4096.  Off the hot path: one wave
 * per query row, fp32 arithmetic (the ViT block uses chb_attention_fwd / _bwd).  Backward: dq fp32 [B*Tq, H*hd] written; dk, dv fp32
 * [B*Tk, H*hd] ACCUMULATED with atomics (the caller zeroes them). */
int chb_attention_general_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo,
                              float* lse, int B, int Tq, int Tk, int H, int hd, const uint8_t* value_mask, const uint8_t* query_mask,
                              int causal, float drop_rate, uint32_t drop_key, float scale, void* stream);
int chb_attention_general_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o, int64_t ldo,
                              const void* d_o, int64_t ldg, const float* lse, float* dq, float* dk, float* dv, int B, int Tq, int Tk, int H,
                              int hd, const uint8_t* value_mask, const uint8_t* query_mask, int causal, float drop_rate, uint32_t drop_key,
                              float scale, void* stream);
/* scale: the factor on the raw scores, 1 / sqrt(key_dim) when ScaledAttention was given a key_dim (layers/attention.py:8-22 divides by
 * sqrt(key_dim) whatever the width of the tensors); <= 0 means 1 / sqrt(hd), the key_dim=None case. */

/* ---------------------------------------------------------------- input side */
#define CHB_DT_U8 0
#define CHB_DT_F32 1
#define CHB_RESIZE_BILINEAR 0
#define CHB_RESIZE_NEAREST 1
/* tf.image.resize as keras `Resizing` and chambers `ResizingMinMax` call it (augmentations/__init__.py:11,
 * image_augmentations.py:686-748): NHWC in (uint8 or fp32) -> [B,OH,OW,C]; bilinear writes fp32, nearest the input
 * dtype; half-pixel centres, no antialias (TF2). */
int chb_resize(const void* in, int in_dtype, void* out, int B, int H, int W, int C, int OH, int OW, int method, void* stream);
/* The decode -> Resizing step in front of batching (data/dataset.py:264-315 maps read_and_decode_image, the training scripts map
 * Resizing per element, test_units/data/test_dataset.py:176): B decoded RGB uint8 images of DIFFERENT sizes packed back to back in
 * one device buffer (offsets_dev[b] = byte offset, hw_dev[2b], hw_dev[2b+1] = height, width) -> one [B,OH,OW,3] batch in a single
 * launch.  Same arithmetic as chb_resize per image; out fp32 (tf.image.resize) or uint8 (its tf.cast: truncation).  OW % 4 == 0. */
int chb_resize_ragged(const void* packed_u8, int64_t packed_bytes, const int64_t* offsets_dev, const int32_t* hw_dev, int B, void* out,
                      int out_dtype, int OH, int OW, int method, void* stream);
/* CenterCrop / RandomCrop / RandomFlip (augmentations/__init__.py:1-13) as one gather of whole pixels of
 * `pixel_bytes` bytes: window offset (y, x) = offsets_dev[b] (per_image) / offsets_dev[0] / (oy0, ox0) when NULL;
 * flips_dev[b] bit 0 = left-right, bit 1 = up-down inside the window (NULL = none). */
int chb_crop_flip(const void* in, void* out, int B, int H, int W, int pixel_bytes, int OH, int OW, const int32_t* offsets_dev,
                  int per_image, int oy0, int ox0, const uint8_t* flips_dev, void* stream);
/* keras `Rescaling`: out fp32 [n] = float(in) * scale + offset. */
int chb_rescale(const void* in, int in_dtype, float* out, int64_t n, float scale, float offset, void* stream);

/* ---------------------------------------------------------------- glue around the block */
/* x[b,0,:] = dropout(cls + pos[0]) (ConcatEmbedding + LearnedEmbedding1D + Dropout,
 * layers/embedding.py:179-180,251-261; vision_transformer.py:249-261). x fp32 [B,N,D]. */
int chb_cls_row(float* x, const float* cls, const float* pos, int B, int N, int D, float drop_rate, uint32_t drop_key,
                void* stream);
/* the same for any special-token row: x[b,row,:] = dropout(tok + pos[row]) — row 1 is the distillation token of
 * DistilledVisionTransformer (ConcatEmbedding "add_dist_token", vision_transformer.py:340-347). */
int chb_token_row(float* x, const float* tok, const float* pos, int B, int N, int D, int row, float drop_rate,
                  uint32_t drop_key, void* stream);
/* backward of embedding stage: dx fp32 [B,N,D] -> dpatch bf16 [B*(N-1), D] (masked), dpos fp32
 * [N,D] and dcls fp32 [D] (accumulated, caller zeroes). */
int chb_embed_bwd(const float* dx, void* dpatch_bf16, float* dpos, float* dcls, int B, int N, int D, float drop_rate,
                  uint32_t drop_key, void* stream);
/* the same with n_special tokens in front of the patches: dtok fp32 [n_special, D] (row 0 class, row 1 distillation
 * token), dpatch bf16 [B*(N-n_special), D]. */
int chb_embed_bwd_tokens(const float* dx, void* dpatch_bf16, float* dpos, float* dtok, int B, int N, int D, int n_special,
                         float drop_rate, uint32_t drop_key, void* stream);
/* dz = dy * keep * 1/(1-rate) as bf16 (backward of keras Dropout ahead of a GEMM). */
int chb_dropout_bwd_bf16(const float* dy, int64_t ld, void* dz_bf16, int M, int N, float drop_rate, uint32_t drop_key,
                         void* stream);
/* out[N] += column sums of bf16 x[M,ld] (bias gradients). */
int chb_colsum_bf16(const void* x, int64_t ld, float* out, int M, int N, void* stream);
/* sparse softmax cross-entropy from logits (mean over batch) + gradient:
 * loss_per_sample fp32 [B]; dlogits bf16 [B,ld_d] = (softmax - onehot) * grad_scale, pad cols 0.
 * Labels are range-checked on the device: a label outside [0, classes) gives loss = NaN and a NaN gradient row
 * (never an out-of-bounds read of the logits). */
int chb_softmax_ce(const float* logits, int64_t ld, const int32_t* labels, float* loss_per_sample, void* dlogits_bf16,
                   int64_t ld_d, int B, int classes, float grad_scale, void* stream);
/* pooling="cls" needs no kernel: chb_layernorm_* take the row stride N*D and touch the cls rows only.
 * pooling="avg" | "max" | "sum" (vision_transformer.py:172-181: Cropping1D((1,0)) drops the cls token, then
 * GlobalAveragePooling1D / GlobalMaxPooling1D / Sum over the remaining N-1 tokens):
 * h bf16 [B,N,D] (final LayerNorm output) -> out bf16 [B,D] (fp32 reduction); mode CHB_POOL_*.
 * argmax int32 [B,D] (token index of the first maximum) is written for CHB_POOL_MAX and may be NULL otherwise. */
#define CHB_POOL_AVG 0
#define CHB_POOL_MAX 1
#define CHB_POOL_SUM 2
int chb_pool_tokens(const void* h_bf16, void* out_bf16, int32_t* argmax, int B, int N, int D, int mode, void* stream);
/* backward: dout bf16 [B,D] -> dh bf16 [B,N,D]; row 0 (cls) gets zeros; avg: dout/(N-1) on every token, sum: dout,
 * max: dout on the argmax token of each (b, column), zero elsewhere (ties: first maximum). */
int chb_pool_tokens_bwd(const void* dout_bf16, const int32_t* argmax, void* dh_bf16, int B, int N, int D, int mode,
                        void* stream);
/* Dense(feature_dim, activation="tanh", name="feature") (vision_transformer.py:275-278) around chb_gemm_nt:
 * forward  y = tanh(z) in place on fp32 [n] plus a bf16 copy (operand of the next GEMM);
 * backward dz bf16 [n] = dy * (1 - y*y), dy / y fp32. */
int chb_tanh_fwd(float* z_inout, void* y_bf16, int64_t n, void* stream);
int chb_tanh_bwd(const float* dy, const float* y, void* dz_bf16, int64_t n, void* stream);

/* ---------------------------------------------------------------- metric-learning head */
/* L2Normalization(axis=-1) (layers/normalization.py:5-24; tf.nn.l2_normalize): y = x * rsqrt(max(sum x^2, 1e-12)),
 * inv_norm fp32 [B] saved for backward; bwd: dx = inv * (dy - y <y, dy>). */
int chb_l2_normalize_fwd(const float* x, float* y, float* inv_norm, int B, int D, void* stream);
int chb_l2_normalize_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int B, int D, void* stream);
/* MultiSimilarityLoss + MultiSimilarityMiner (losses/metric_learning.py:124-178, miners.py:48-60) on embeddings fp32
 * [B,D] and int32 labels [B]: loss_rows[i] as in compute_loss (the Keras loss value is their mean); when d_emb is not
 * NULL also d(mean loss)/d(emb) fp32 [B,D], using workspace fp32 [B,B].  use_miner = 0 reproduces miner=None. */
int chb_multi_similarity_loss(const float* emb, const int32_t* labels, float* loss_rows, float* workspace, float* d_emb,
                              int B, int D, float pos_scale, float neg_scale, float threshold, float miner_margin,
                              int use_miner, int ignore_diag, int ignore_negative_labels, void* stream);
/* The PairMatrixLoss form (losses/metric_learning.py:112-121, MultiSimilarityLossMatrix :181-235): `sim` fp32 [B,B] IS the similarity
 * matrix, positive_mask uint8 [B,B] its boolean positive-pair mask (y_true cast to bool); d_sim (NULL or fp32 [B,B]) receives
 * d(mean loss)/d(sim). */
int chb_multi_similarity_loss_matrix(const float* sim, const uint8_t* positive_mask, float* loss_rows, float* d_sim, int B, float pos_scale,
                                     float neg_scale, float threshold, float miner_margin, int use_miner, int ignore_diag, void* stream);
/* ContrastiveLoss (losses/metric_learning.py:238-287) on embeddings: loss_rows[i] = sum over kept positives of (positive_margin - s)^e / e
 * + sum over kept negatives of max(0, s - negative_margin)^e / e (tf.pow semantics, e = exponent); pairs, miner, gradient and
 * workspace as for chb_multi_similarity_loss. */
int chb_contrastive_loss(const float* emb, const int32_t* labels, float* loss_rows, float* workspace, float* d_emb, int B, int D,
                         float positive_margin, float negative_margin, float exponent, float miner_margin, int use_miner, int ignore_diag,
                         int ignore_negative_labels, void* stream);
/* NTXentLoss (losses/metric_learning.py:290-323): Keras CategoricalCrossentropy of the rows of emb.emb^T / temperature (diagonal set to
 * -1e9) against the multi-hot rows [label_j == label_i, j != i]; from_logits != 0: -sum y log softmax(z); else the probability form
 * (rows divided by their sum, clipped to [1e-7, 1 - 1e-7]).  loss_rows fp32 [B] (the Keras value is their mean), d_emb / workspace as above. */
int chb_ntxent_loss(const float* emb, const int32_t* labels, float* loss_rows, float* workspace, float* d_emb, int B, int D, float temperature,
                    int from_logits, void* stream);

/* fp32 [R,C] -> bf16 [R,C] and/or bf16 [C,R] for a table of matrices (one launch). desc is a
 * device int64 array [n,4] = {src_offset, dst_offset, R, C} in elements; dst_t gets the
 * transposed copy at dst_offset (same element offsets). Either dst may be NULL. */
int chb_cast_transpose(const float* src, void* dst_bf16, void* dst_t_bf16, const int64_t* desc, int n_desc,
                       int max_tiles, void* stream);

/* AdamW (optimizers.py:147-155,372-464 + keras Adam): per element, if decay flag of its
 * 1024-element chunk is set: p -= wd*p; then m += (g-m)(1-b1); v += (g*g-v)(1-b2);
 * p -= lr_t*m/(sqrt(v)+eps).  grad_scale multiplies g first (1/world for data parallel).
 * zero_grad != 0: g[i] = 0 right behind its read, so the next backward accumulates into zeros without a fill pass
 * (what keras' apply_gradients + a fresh GradientTape amount to). */
int chb_adamw(float* p, float* g, float* m, float* v, const uint8_t* decay_flags, int64_t n, float lr_t,
              float beta1, float beta2, float eps, float weight_decay, float grad_scale, int zero_grad, void* stream);
/* RandomChoice(elementwise=True) (image_augmentations.py:563-570: tf.map_fn of `_random_transforms` over batch-1 tensors,
 * :606-617) and the elementwise modes of RandAugment / AutoAugment (augmentation_schemes.py:138-149,193): ONE slot of the
 * scheme for the whole batch, every image with its own op and parameters.  items_dev: B records of 64 bytes on the device,
 *   struct { int32 op; int32 i0, i1, i2, i3; int32 pad[3]; float f[8]; }   (op = CHB_AUG_*, fields as listed there),
 * written by the host from the per-image decisions (op index, sign draw, cutout centre; Contrast's constant is that of a
 * batch-1 tensor).  n_stats = number of images whose op is AutoContrast / Equalize (0 skips the histogram and table
 * launches); workspace int32 [B*3*256] is needed when n_stats > 0.  uint8 NHWC RGB, in != out.  Results equal the
 * batch ops above applied image by image, bit for bit. */
int chb_aug_dispatch(const uint8_t* in, uint8_t* out, int B, int H, int W, const void* items_dev, int n_stats,
                     int32_t* workspace, void* stream);

/* Fused scheme stage for BATCH-SHARED decisions (RandAugment.call augmentation_schemes.py:204-213, AutoAugment :151-160; then
 * ImageNetNormalization("tf") image_augmentations.py:659-665 and the patch gather of vision_transformer.py:235-248): the chain
 * of n_ops <= CHB_FUSED_MAX_OPS ops is evaluated per output pixel in ONE pass over the batch.  ops_host: n_ops HOST records
 *   struct { int32 op; int32 i0, i1, i2, i3; float f[6]; int32 pad; }    (48 bytes; op = CHB_AUG_*, fields as for chb_aug_dispatch;
 *   Contrast's i0 is the constant of the whole batch tensor, B*H*W/256 clipped), applied in order.
 * centers_dev[l]: device int32 [B,2] (cy, cx) for a CutOut at level l (else ignored / NULL).  workspace: int32
 * [chb_aug_fused_workspace_ints(B, H, W, n_tables)], uninitialised, when the chain holds AutoContrast / Equalize (each costs a
 * histogram pass of the level below it, which leaves one partial table per workgroup, + a table launch that adds them up).
 * scratch: NULL, or 2*B*H*W*3 bytes.  NULL: the whole chain is evaluated per output pixel, the levels under a Sharpness at
 * each of its nine taps.  With scratch the chain is cut at its Sharpness ops, which then read a materialised uint8 image
 * through the stand-alone kernel (same bytes out, fewer gathers; the faster route); everything between two cuts is one launch.
 * patch == 0: out = uint8 NHWC [B,H,W,3] (the chain's image);  patch > 0 (multiple of 4): out = bf16 [B*(H/patch)*(W/patch),
 * patch*patch*3] patch rows of the "tf"-normalised chain output (what chb_normalize_patchify_bf16 would produce from it).
 * Bit-identical to running the ops one after the other. */
#define CHB_FUSED_MAX_OPS 4
int chb_aug_fused(const uint8_t* in, void* out, int B, int H, int W, int n_ops, const void* ops_host,
                  const int32_t* const* centers_dev, int32_t* workspace, uint8_t* scratch, int patch, void* stream);
/* The same stage for the schemes' elementwise=True mode (image_augmentations.py:563-570, augmentation_schemes.py:135: tf.map_fn over
 * batch-1 tensors - every image draws its own ops, signs, cutout centre, and Contrast's constant is its own H*W/256).  items_dev:
 * DEVICE records, one per (level, image), [n_ops][B], laid out as chb_aug_fused's host records.  centers_dev[l]: device int32 [B,2]
 * for the images with a CutOut at level l (NULL if none; a CutOut record at a level without a table leaves its image untouched).  table_levels: bit l set if some image has an AutoContrast / Equalize at
 * level l; workspace: chb_aug_fused_workspace_ints(B, H, W, popcount(table_levels)) int32, uninitialised.  One launch for the batch
 * (+ a histogram pass and a table launch per set bit); out as for chb_aug_fused.  Bit-identical to the ops applied image by image. */
int chb_aug_fused_items(const uint8_t* in, void* out, int B, int H, int W, int n_ops, const void* items_dev,
                        const int32_t* const* centers_dev, int table_levels, int32_t* workspace, int patch, void* stream);
/* The elementwise stage with the images SORTED by what their chains need, so that every group runs the evaluator the batch-shared
 * launches would pick for it (same reference lines as chb_aug_fused_items; the reference maps the images one by one, any order of
 * evaluation gives its bytes).  chb_aug_items_sort (host only): recs_host = the [n_ops][B] records before upload - the `pad` member of
 * warps that are pure row shifts on H x W is set in place; order_out int32 [1 + n_ops][B]: row 0 = every image, in six groups - the
 * chains WITHOUT an AutoContrast / Equalize by kind (pixel-local | no Sharpness and only warps that keep rows | the rest, chains with a
 * Sharpness first), then the chains with one by kind; row 1 + l = the images with an AutoContrast / Equalize at level l by the kind of
 * the levels under it (groups 0 .. 2); counts_out int32 [1 + n_ops][CHB_ITEMS_GROUPS] = the group sizes.
 * chb_aug_fused_items_sorted: order_dev = that array in device memory, counts_host = the counts; everything else as
 * chb_aug_fused_items.  One final launch per non-empty group and a histogram launch per (table level, non-empty kind) over the images
 * that need one; the chains without tables run beside the histogram passes (the groups are independent: caller's stream + internal
 * side streams, forked and joined by events inside the call, capturable).  Same bytes as chb_aug_fused_items. */
#define CHB_ITEMS_GROUPS 6
int chb_aug_items_sort(void* recs_host, int B, int H, int W, int n_ops, int32_t* order_out, int32_t* counts_out);
int chb_aug_fused_items_sorted(const uint8_t* in, void* out, int B, int H, int W, int n_ops, const void* items_dev,
                               const int32_t* const* centers_dev, int table_levels, int32_t* workspace, int patch,
                               const int32_t* order_dev, const int32_t* counts_host, void* stream);
/* int32 elements of chb_aug_fused's workspace for a chain with n_tables AutoContrast / Equalize ops (0 for none): host arithmetic only. */
int64_t chb_aug_fused_workspace_ints(int B, int H, int W, int n_tables);

/* Tuning / A-B switch `name` (ATTN_FWD_ALGO, ATTN_BWD_ALGO, AFFINE_ALGO, GEMM_ALGO, GEMM_WALK, TN_ATOMICS, TN_FAST,
 * GEMM_TILE_QUEUE, LN_STREAM; csrc/common.hpp) := value.  Defaults come from the environment variables CHB_<name>, read once per
 * process; nothing on the launch path calls getenv.  Results are identical under every setting (the parity tests
 * cross-check them); only speed changes. */
int chb_set_option(const char* name, int value);
/* Zero the tile-queue counters of the persistent NT GEMMs (GEMM_TILE_QUEUE = 1; default 0 = static tile shares, no counters used):
 * the only device state the library keeps between launches.  Launches clean up behind themselves; this is for the start of a run
 * and for recovery after a faulted launch.  Ordered on `stream`. */
int chb_gemm_tile_queue_reset(void* stream);
/* x[0..n) = 0 (n % 4 == 0, 16-byte aligned): gradient buffer reset when backward runs twice without an optimizer step. */
int chb_zero_f32(float* x, int64_t n, void* stream);

/* Small tensor utilities of the stand-alone Keras-style layers (torch allocates, this library computes):
 * out = a + b (fp32, 16-byte aligned) - the first residual of EncoderLayer.call (layers/transformer.py:72-74);
 * fp32 <-> bf16 casts of GEMM operands; a strided 2-D copy (rows x row_bytes; all strides, sizes and pointers multiples of 4) -
 * ConcatEmbedding's tf.concat (layers/embedding.py:100-104); softmax over the last axis of fp32 rows - the classifier_activation
 * of the stand-alone Dense / of model.predict (vision_transformer.py:283). */
int chb_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);
int chb_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
int chb_cast_bf16_f32(const void* src, float* dst, int64_t n, void* stream);
int chb_copy_rows(const void* src, int64_t src_stride_bytes, void* dst, int64_t dst_stride_bytes, int64_t rows, int64_t row_bytes, void* stream);
int chb_softmax_f32(const float* x, int64_t ld, float* out, int64_t ld_out, int rows, int cols, void* stream);
/* The two outputs of DistilledVisionTransformer (vision_transformer.py:373-397: class-token head and distillation-token head, a pair or
 * their average): out = alpha * a + beta * b (fp32; b may be NULL: out = alpha * a) - the average and its backward;
 * x[r][c] = bf16(x[r][c] + y[r][c]) over strided bf16 rows (ld in elements) - the distillation head's gradient joining row 1 of the
 * normalised sequence. */
int chb_axpby_f32(const float* a, float alpha, const float* b, float beta, float* out, int64_t n, void* stream);
int chb_add_rows_bf16(void* x_bf16, int64_t ldx, const void* y_bf16, int64_t ldy, int64_t rows, int cols, void* stream);

/* Pieces of the TRAINABLE stand-alone layers (chambers_amd/layers: torch.autograd.Function wrappers over this ABI; the whole-model
 * engine has these fused into its GEMM / LayerNorm epilogues):
 * chb_gelu_f32: y = gelu(x), dydx (optional) = gelu'(x); approximate = 0 the exact-erf form, 1 the tanh form (activations.py:5-56).
 * chb_mul_f32: out = a * b (backward of the stand-alone activation).
 * chb_scale_by_bf16: out_bf16 = bf16(dy * aux_bf16), dy fp32 (CHB_OUT_F32) or bf16 (CHB_OUT_BF16) - backward of Dense(activation=gelu)
 *   from the gelu' its forward GEMM epilogue saved (layers/transformer.py:42-44).
 * chb_dropout_f32: out = x * keep / (1 - rate), keep = the chb_dropout_mask definition on the FLAT element index - keras Dropout as a
 *   layer (layers/transformer.py:38,48; vision_transformer.py:261), forward and backward alike.
 * chb_add_rows_f32: out[i] = x[i] + table[i mod period] - LearnedEmbedding1D (layers/embedding.py:156-182), period = tokens * dim.
 * chb_sum_rows_f32: out[c] = sum_r x[r * row_stride + c] - the batch reduction in the backward of LearnedEmbedding1D / ConcatEmbedding
 *   (layers/embedding.py:218-261). */
int chb_gelu_f32(const float* x, float* y, float* dydx, int64_t n, int approximate, void* stream);
int chb_mul_f32(const float* a, const float* b, float* out, int64_t n, void* stream);
int chb_scale_by_bf16(const void* dy, int dy_dtype, const void* aux_bf16, void* out_bf16, int64_t n, void* stream);
int chb_dropout_f32(const float* x, float* out, int64_t n, float rate, uint32_t key, void* stream);
int chb_add_rows_f32(const float* x, const float* table, float* out, int64_t n, int64_t period, void* stream);
int chb_sum_rows_f32(const float* x, int64_t row_stride, int64_t rows, int64_t cols, float* out, void* stream);

/* ---- one encoder block per call -------------------------------------------------------------------------------------------------
 * EncoderLayer.call, pre-norm branch (layers/transformer.py:53-77; the block sequence of models/backbones/vision_transformer.py:
 * 262-271) and its gradient as ONE host call each: chb_vit_block_fwd / _bwd issue, on `stream`, exactly the chb_layernorm_* /
 * chb_gemm_* / chb_attention_* launches listed above with the arguments this record implies (same kernels, same order: results are
 * bit-identical to driving those entries one by one); the caller fills the record once per block (all buffers are the caller's and
 * static) and only updates the dropout site keys per step.  Nothing is allocated; no state is kept between calls except the events
 * of the optional side stream.
 *   token matrices have Mp rows (multiple of 256, >= M = B*N); GEMMs are launched over Mg rows (M, or Mp: pad rows hold finite junk
 *   in the forward and exact zeros in the backward), reductions over tokens run over M.
 *   bf16 matrices: *_wt = [out][in] images (forward B operands), *_w = [in][out] images (dgrad B operands), both refreshed by
 *   chb_cast_transpose after every optimizer step; g_*_w fp32 [in][out] gradient accumulators (+=).
 *   backward contract: on entry dx (fp32 [Mp,D]) holds d(loss)/d(x_out) and dz (bf16 [Mp,D]) its dropout-backward at this block's MLP
 *   site (key_mlp) - written by the block above or by the final LayerNorm's fused tail; on exit dx holds d(loss)/d(x_in) and, with
 *   emit_dz != 0, dz its dropout-backward at the MLP site of the block BELOW (key_prev_mlp) with the column sums added to
 *   g_prev_fc2_bias (that block's dense2 bias gradient).
 *   phases (backward): bit 0 = MLP branch + output projection, bit 1 = attention core + QKV projection + LayerNorm 1.  A data-parallel
 *   caller runs phase 1, starts the all-reduce of the gradients that are now final, then phase 2 (engine.py GradBucketReducer);
 *   phases = 3 runs the whole block.
 *   side_stream (backward, optional): the four weight-gradient GEMMs go to this second stream (fork / join by events inside the
 *   call; they depend only on saved activations and on dz / da1 / dqkv, and only the optimizer reads their output), using tn_ws_side
 *   as their split-K scratch; chb_side_stream_join makes `stream` wait for everything issued there (call it before the optimizer or
 *   a collective reads the weight gradients).  NULL: everything on `stream`. */
typedef struct chb_vit_block {
    int32_t B, N, H, hd, D, FF;             /* images, tokens per image, heads, head dim (64), model width, MLP width */
    int32_t M, Mg, Mp;                      /* B*N token rows; rows the GEMMs are launched over; allocated rows */
    float eps, drop_rate;                   /* LayerNorm epsilon (1e-6, layers/transformer.py:20); dropout rate of the three sites */
    uint32_t key_attn, key_proj, key_mlp;   /* dropout site keys of this block for this step */
    uint32_t key_prev_mlp;                  /* backward: MLP site key of the block below */
    int32_t emit_dz;                        /* backward: 0 for the lowest block (the embedding stage takes dx) */
    int32_t reserved;
    const float *ln1_gamma, *ln1_beta, *ln2_gamma, *ln2_beta, *qkv_bias, *proj_bias, *fc1_bias, *fc2_bias;
    const void *qkv_wt, *proj_wt, *fc1_wt, *fc2_wt;
    const void *qkv_w, *proj_w, *fc1_w, *fc2_w;
    const float* x_in;                      /* fp32 [Mp,D] block input (saved: LayerNorm 1 backward reads it) */
    float* x_out;                           /* fp32 [Mp,D] */
    void *h1, *qkv, *o, *h2, *a1, *u;       /* bf16 [Mp,D], [Mp,3D], [Mp,D], [Mp,D], [Mp,FF] (gelu'), [Mp,FF] (gelu) */
    float *mean1, *rstd1, *lse, *xmid, *mean2, *rstd2;   /* fp32 [Mp], [Mp], [B*H*N], [Mp,D], [Mp], [Mp] */
    uint32_t* drop_bits;                    /* keep bits of the attention dropout (chb_attention_fwd), or NULL */
    float *g_ln1_gamma, *g_ln1_beta, *g_ln2_gamma, *g_ln2_beta, *g_qkv_bias, *g_proj_bias, *g_fc1_bias, *g_prev_fc2_bias;
    float *g_qkv_w, *g_proj_w, *g_fc1_w, *g_fc2_w;
    float* dx;                              /* fp32 [Mp,D] residual-stream gradient, in / out */
    void *dz, *da1, *dh, *d_o, *dqkv;       /* bf16 scratch [Mp,D], [Mp,FF], [Mp,D], [Mp,D], [Mp,3D]; dz in / out */
    float *tn_ws, *tn_ws_side;              /* split-K scratch of the weight-gradient GEMMs (chb_gemm_tn_ws), tn_ws_bytes each */
    int64_t tn_ws_bytes;
    float* tn_ws4;                          /* optional: 4 x tn_ws_bytes - one scratch per weight gradient of the block, folded by ONE launch per
                                               chb_vit_block_bwd call (chb_gemm_tn_fold_multi) instead of one per GEMM; NULL: fold per GEMM */
} chb_vit_block;
int chb_vit_block_fwd(const chb_vit_block* block_host, int training, void* stream);
int chb_vit_block_bwd(const chb_vit_block* block_host, int phases, void* stream, void* side_stream);
int chb_side_stream_join(void* stream, void* side_stream);

/* ---- launch profiler -----------------------------------------------------------------------------------------------------------
 * chb_profile_enable(1) starts a period (dropping earlier records): from then on chb_gemm_nt and chb_gemm_tn / chb_gemm_tn_ws -
 * called directly or from chb_vit_block_* - record a HIP event on their launch stream right before and right after the GEMM launch
 * (for the weight gradients: the GEMM alone, not the fold of its split-K planes).  chb_profile_enable(0) stops recording.
 * chb_profile_collect waits for the recorded events and copies up to max_records records to HOST memory; *n_records_host = how many
 * exist.  kind 0 = chb_gemm_nt (family 1: 128x128 tiles, 2 / 4 / 5: the persistent 256x256 kernels - lockstep, ping-pong, pipelined),
 * kind 1 = weight gradient (family 0: small, 1: persistent 256x256); m, n, k as passed (k = reduction length);
 * ms = launch duration, start_ms = its start relative to the period's first record (launches of two streams may overlap).
 * Recording is skipped while a stream is being captured into a graph.  Host-side bookkeeping only: no device state. */
typedef struct chb_profile_record {
    int32_t kind, family, epilogue, out_dtype;
    int64_t m, n, k;
    float ms, start_ms;
} chb_profile_record;
int chb_profile_enable(int on);
int chb_profile_collect(chb_profile_record* out_host, int max_records, int* n_records_host);

#ifdef __cplusplus
}
#endif
#endif /* CHAMBERS_HIP_H */
