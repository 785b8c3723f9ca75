// Metric-learning head of the ViT backbones: L2Normalization (layers/normalization.py:5-24) and
// MultiSimilarityLoss with the MultiSimilarityMiner (losses/metric_learning.py:9-178, miners.py:48-60) on the [B, D] embeddings
// the `feature` head produces.  B is a batch (<= a few thousand), so everything is small: one workgroup per anchor row.
//   similarity s_ij = <f_i, f_j> (fp32); positives = same label, negatives = different label, optionally without the diagonal and
//   without anchors / pairs whose label is negative; miner: keep positives with s < max_neg_i + margin and negatives with
//   s > min_pos_i - margin (empty sets reduce to -FLT_MAX / +FLT_MAX like tf.reduce_max / reduce_min on empty ragged rows);
//   loss_i = log(1 + sum_pos exp(-a (s - l))) / a + log(1 + sum_neg exp(b (s - l))) / b; the Keras loss is the batch mean.
// Backward: w_ij = d(mean loss)/d(s_ij) (mining masks are constants), df_i = sum_j (w_ij + w_ji) f_j.
// Round 3: the other pair losses of the file - ContrastiveLoss (:238-287: sum_pos (pm - s)^e / e + sum_neg max(0, s - nm)^e / e), the
// PairMatrixLoss form (:112-121, MultiSimilarityLossMatrix :181-235: y_pred IS the similarity matrix, y_true its boolean positive
// mask; the gradient is the pair-weight matrix itself) and NTXentLoss (:290-323: Keras CategoricalCrossentropy of the temperature-
// scaled similarity rows with the diagonal at -1e9 against the multi-hot same-label rows, from_logits or not).
#include "common.hpp"
#include "../../include/chambers_hip.h"
#include <float.h>

namespace {

__device__ __forceinline__ float block_reduce(float v, float* red, int op) {   // op 0 sum, 1 max, 2 min; 256 threads
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(v, o, 64);
        v = op == 0 ? v + t : (op == 1 ? fmaxf(v, t) : fminf(v, t));
    }
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int k = 1; k < 4; ++k) r = op == 0 ? r + red[k] : (op == 1 ? fmaxf(r, red[k]) : fminf(r, red[k]));
    return r;
}

// y = x / sqrt(max(sum x^2, 1e-12)) (tf.nn.l2_normalize); inv = that reciprocal root
__global__ void __launch_bounds__(256) l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv, int D) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) { const float v = x[(int64_t)row * D + d]; s += v * v; }
    s = block_reduce(s, red, 0);
    const float r = rsqrtf(fmaxf(s, 1e-12f));
    for (int d = threadIdx.x; d < D; d += 256) y[(int64_t)row * D + d] = x[(int64_t)row * D + d] * r;
    if (threadIdx.x == 0) inv[row] = r;
}

// dx = inv * (dy - y <y, dy>)   (rows whose squared norm was clamped pass dy * inv through: d(clamp) = 0)
__global__ void __launch_bounds__(256) l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ inv,
                                                         float* __restrict__ dx, int D) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) s += y[(int64_t)row * D + d] * dy[(int64_t)row * D + d];
    s = block_reduce(s, red, 0);
    const float r = inv[row];
    const float proj = (r >= 1e6f) ? 0.f : s;     // inv == 1e6 <=> the squared norm hit the 1e-12 floor
    for (int d = threadIdx.x; d < D; d += 256) dx[(int64_t)row * D + d] = r * (dy[(int64_t)row * D + d] - y[(int64_t)row * D + d] * proj);
}

struct PairParams {
    int kind;            // 0 MultiSimilarity, 1 Contrastive
    float alpha, beta, lambda, margin;       // MultiSimilarity: pos_scale, neg_scale, threshold; miner margin (both kinds)
    float pos_margin, neg_margin, exponent;  // Contrastive
    int use_miner, ignore_diag, ignore_neg;
};

// one workgroup per anchor i: similarities to every j (computed from the embeddings, or row i of a given matrix), mining thresholds,
// loss_i and w_ij = d(loss_i)/d(s_ij) * grad_scale.  MATRIX: `f` is the [B,B] similarity matrix, `labels` a uint8 [B,B] positive mask.
template <bool MATRIX>
__global__ void __launch_bounds__(256) pair_rows_kernel(const float* __restrict__ f, const void* __restrict__ labels, float* __restrict__ loss_rows,
                                                        float* __restrict__ wmat, int B, int D, PairParams p, float grad_scale) {
    extern __shared__ float sim[];     // [B]
    __shared__ float red[4];
    const int i = blockIdx.x;
    const int32_t* lab = reinterpret_cast<const int32_t*>(labels);
    const uint8_t* msk = reinterpret_cast<const uint8_t*>(labels) + (int64_t)i * B;
    const int li = MATRIX ? 0 : lab[i];
    const float* fi = f + (int64_t)i * (MATRIX ? B : D);
    const bool row_valid = MATRIX || !(p.ignore_neg && li < 0);       // (a boolean mask is never negative: :112-121 after :87-90)
    float mx_neg = -FLT_MAX, mn_pos = FLT_MAX;
    for (int j = threadIdx.x; j < B; j += 256) {
        float s = 0.f;
        if (MATRIX) {
            s = fi[j];
        } else {
            const float* fj = f + (int64_t)j * D;
            for (int d = 0; d < D; ++d) s += fi[d] * fj[d];
        }
        sim[j] = s;
        const bool same = MATRIX ? (msk[j] != 0) : (lab[j] == li);
        const bool valid = !(p.ignore_diag && j == i) && row_valid;
        if (valid) {
            if (same) mn_pos = fminf(mn_pos, s);
            else mx_neg = fmaxf(mx_neg, s);
        }
    }
    mx_neg = block_reduce(mx_neg, red, 1);
    mn_pos = block_reduce(mn_pos, red, 2);
    const float pos_thr = mx_neg + p.margin, neg_thr = mn_pos - p.margin;
    float sp = 0.f, sn = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) {
        const float s = sim[j];
        const bool same = MATRIX ? (msk[j] != 0) : (lab[j] == li);
        const bool valid = !(p.ignore_diag && j == i) && row_valid;
        const bool kp = valid && same && (!p.use_miner || s < pos_thr), kn = valid && !same && (!p.use_miner || s > neg_thr);
        float e = 0.f;
        if (p.kind == 0) {
            // signed exponential: negative for a kept positive pair, positive for a kept negative pair
            if (kp) { e = expf(-p.alpha * (s - p.lambda)); sp += e; e = -e; }
            else if (kn) { e = expf(p.beta * (s - p.lambda)); sn += e; }
        } else {
            // the pair's loss term goes into the row sum, its derivative (tf.pow: x^e -> e x^(e-1), /e) into sim[]
            if (kp) { const float x = p.pos_margin - s; sp += powf(x, p.exponent) / p.exponent; e = -powf(x, p.exponent - 1.0f); }
            else if (kn) {
                const float x = fmaxf(0.0f, s - p.neg_margin);
                sn += powf(x, p.exponent) / p.exponent;
                e = (s - p.neg_margin > 0.0f) ? powf(x, p.exponent - 1.0f) : 0.0f;      // d max(0, .) = 0 at and below the margin
            }
        }
        sim[j] = e;
    }
    sp = block_reduce(sp, red, 0);
    sn = block_reduce(sn, red, 0);
    if (threadIdx.x == 0) loss_rows[i] = p.kind == 0 ? logf(1.0f + sp) / p.alpha + logf(1.0f + sn) / p.beta : sp + sn;
    if (wmat) {
        const float cp = p.kind == 0 ? grad_scale / (1.0f + sp) : grad_scale, cn = p.kind == 0 ? grad_scale / (1.0f + sn) : grad_scale;
        for (int j = threadIdx.x; j < B; j += 256) {
            const float e = sim[j];
            wmat[(int64_t)i * B + j] = e < 0.f ? e * cp : e * cn;     // MS: d/ds of log(1 + sum exp(-a(s-l)))/a = -exp(..)/(1+sum)
        }
    }
}

// NTXentLoss row i: z_j = <f_i, f_j> / T (z_i = -1e9), y_j = [label_j == label_i, j != i]; from_logits: -sum_j y_j log softmax(z)_j;
// otherwise Keras' probability form: o = z / sum(z), clipped to [1e-7, 1 - 1e-7], -sum_j y_j log o_j.  w_ij = d(loss_i)/d(s_ij) * scale.
__global__ void __launch_bounds__(256) ntxent_rows_kernel(const float* __restrict__ f, const int32_t* __restrict__ lab, float* __restrict__ loss_rows,
                                                          float* __restrict__ wmat, int B, int D, float inv_t, int from_logits, float grad_scale) {
    extern __shared__ float sim[];
    __shared__ float red[4];
    const int i = blockIdx.x;
    const int li = lab[i];
    const float* fi = f + (int64_t)i * D;
    float mx = -FLT_MAX, tot = 0.f, ny = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) {
        const float* fj = f + (int64_t)j * D;
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += fi[d] * fj[d];
        s = (j == i) ? -1.0e9f : s * inv_t;
        sim[j] = s;
        mx = fmaxf(mx, s);
        tot += s;
        ny += (j != i && lab[j] == li) ? 1.0f : 0.0f;
    }
    mx = block_reduce(mx, red, 1);
    tot = block_reduce(tot, red, 0);
    ny = block_reduce(ny, red, 0);
    const float eps = 1.0e-7f;
    float acc = 0.f, acc2 = 0.f;       // from_logits: sum exp(z - mx), sum y z;  else: sum y log o, sum y [unclipped] z / (o S^2)
    for (int j = threadIdx.x; j < B; j += 256) {
        const float z = sim[j];
        const bool y = j != i && lab[j] == li;
        if (from_logits) {
            acc += expf(z - mx);
            if (y) acc2 += z;
        } else {
            const float o = z / tot;
            const float c = fminf(fmaxf(o, eps), 1.0f - eps);
            if (y) {
                acc += logf(c);
                if (o > eps && o < 1.0f - eps) acc2 += z / (c * tot * tot);
            }
        }
    }
    acc = block_reduce(acc, red, 0);
    acc2 = block_reduce(acc2, red, 0);
    if (threadIdx.x == 0) loss_rows[i] = from_logits ? ny * (mx + logf(acc)) - acc2 : -acc;
    if (wmat) {
        for (int j = threadIdx.x; j < B; j += 256) {
            const float z = sim[j];
            const bool y = j != i && lab[j] == li;
            float g = 0.f;
            if (j != i) {       // (the diagonal is a constant: tf.linalg.set_diag)
                if (from_logits) {
                    g = ny * expf(z - mx) / acc - (y ? 1.0f : 0.0f);
                } else {
                    const float o = z / tot;
                    const bool open = o > eps && o < 1.0f - eps;
                    g = acc2 - ((y && open) ? 1.0f / (fminf(fmaxf(o, eps), 1.0f - eps) * tot) : 0.0f);
                }
            }
            wmat[(int64_t)i * B + j] = g * inv_t * grad_scale;
        }
    }
}

// df_i = sum_j (w_ij + w_ji) f_j
__global__ void __launch_bounds__(256) ms_grad_kernel(const float* __restrict__ f, const float* __restrict__ wmat, float* __restrict__ df, int B, int D) {
    const int i = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (int j = 0; j < B; ++j) acc += (wmat[(int64_t)i * B + j] + wmat[(int64_t)j * B + i]) * f[(int64_t)j * D + d];
        df[(int64_t)i * D + d] = acc;
    }
}

}  // namespace

extern "C" {

int chb_l2_normalize_fwd(const float* x, float* y, float* inv_norm, int B, int D, void* stream) {
    if (B < 0 || D <= 0) return CHB_EINVAL;
    if (B == 0) return CHB_OK;
    if (!x || !y || !inv_norm) return CHB_EINVAL;
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, y, inv_norm, D);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_l2_normalize_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int B, int D, void* stream) {
    if (B < 0 || D <= 0) return CHB_EINVAL;
    if (B == 0) return CHB_OK;
    if (!dy || !y || !inv_norm || !dx) return CHB_EINVAL;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dy, y, inv_norm, dx, D);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

static int pair_loss_launch(const float* emb, const void* labels, float* loss_rows, float* workspace, float* grad, int B, int D, bool matrix,
                            const PairParams& p, void* stream) {
    if (B == 0) return CHB_OK;
    if (!emb || !labels || !loss_rows) return CHB_EINVAL;
    if (!matrix && grad && !workspace) return CHB_EINVAL;
    if ((size_t)B * sizeof(float) > 60 * 1024) return CHB_EUNSUPPORTED;      // one row of similarities lives in LDS
    hipStream_t s = (hipStream_t)stream;
    const float scale = 1.0f / (float)B;
    if (matrix) {       // the gradient with respect to the similarity matrix is the pair-weight matrix itself
        hipLaunchKernelGGL(pair_rows_kernel<true>, dim3(B), dim3(256), (size_t)B * sizeof(float), s, emb, labels, loss_rows, grad, B, D, p, scale);
    } else {
        hipLaunchKernelGGL(pair_rows_kernel<false>, dim3(B), dim3(256), (size_t)B * sizeof(float), s, emb, labels, loss_rows, grad ? workspace : nullptr,
                           B, D, p, scale);
        if (grad) hipLaunchKernelGGL(ms_grad_kernel, dim3(B), dim3(256), 0, s, emb, workspace, grad, B, D);
    }
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

int chb_multi_similarity_loss(const float* emb, const int32_t* labels, float* loss_rows, float* workspace, float* d_emb, int B, int D,
                              float pos_scale, float neg_scale, float threshold, float miner_margin, int use_miner, int ignore_diag,
                              int ignore_negative_labels, void* stream) {
    if (B < 0 || D <= 0 || pos_scale <= 0.f || neg_scale <= 0.f) return CHB_EINVAL;
    PairParams p{0, pos_scale, neg_scale, threshold, miner_margin, 0.f, 0.f, 0.f, use_miner ? 1 : 0, ignore_diag ? 1 : 0, ignore_negative_labels ? 1 : 0};
    return pair_loss_launch(emb, labels, loss_rows, workspace, d_emb, B, D, false, p, stream);
}

int chb_multi_similarity_loss_matrix(const float* sim, const uint8_t* positive_mask, float* loss_rows, float* d_sim, int B, float pos_scale,
                                     float neg_scale, float threshold, float miner_margin, int use_miner, int ignore_diag, void* stream) {
    if (B < 0 || pos_scale <= 0.f || neg_scale <= 0.f) return CHB_EINVAL;
    PairParams p{0, pos_scale, neg_scale, threshold, miner_margin, 0.f, 0.f, 0.f, use_miner ? 1 : 0, ignore_diag ? 1 : 0, 0};
    return pair_loss_launch(sim, positive_mask, loss_rows, nullptr, d_sim, B, B, true, p, stream);
}

int chb_contrastive_loss(const float* emb, const int32_t* labels, float* loss_rows, float* workspace, float* d_emb, int B, int D,
                         float positive_margin, float negative_margin, float exponent, float miner_margin, int use_miner, int ignore_diag,
                         int ignore_negative_labels, void* stream) {
    if (B < 0 || D <= 0 || exponent == 0.f) return CHB_EINVAL;
    PairParams p{1, 0.f, 0.f, 0.f, miner_margin, positive_margin, negative_margin, exponent, use_miner ? 1 : 0, ignore_diag ? 1 : 0,
                 ignore_negative_labels ? 1 : 0};
    return pair_loss_launch(emb, labels, loss_rows, workspace, d_emb, B, D, false, p, stream);
}

int chb_ntxent_loss(const float* emb, const int32_t* labels, float* loss_rows, float* workspace, float* d_emb, int B, int D, float temperature,
                    int from_logits, void* stream) {
    if (B < 0 || D <= 0 || temperature == 0.f) return CHB_EINVAL;
    if (B == 0) return CHB_OK;
    if (!emb || !labels || !loss_rows || (d_emb && !workspace)) return CHB_EINVAL;
    if ((size_t)B * sizeof(float) > 60 * 1024) return CHB_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ntxent_rows_kernel, dim3(B), dim3(256), (size_t)B * sizeof(float), s, emb, labels, loss_rows, d_emb ? workspace : nullptr, B, D,
                       1.0f / temperature, from_logits ? 1 : 0, 1.0f / (float)B);
    if (d_emb) hipLaunchKernelGGL(ms_grad_kernel, dim3(B), dim3(256), 0, s, emb, workspace, d_emb, B, D);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

}  // extern "C"
