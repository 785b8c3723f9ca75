// Metric-learning head of the ViT backbones: L2Normalization (layers/normalization.py:5-24) and
// MultiSimilarityLoss with the MultiSimilarityMiner (losses/metric_learning.py:9-178, miners.py:48-60) on the [B, D] embeddings
// the `feature` head produces.  B is a batch (<= a few thousand), so everything is small: one workgroup per anchor row.
//   similarity s_ij = <f_i, f_j> (fp32); positives = same label, negatives = different label, optionally without the diagonal and
//   without anchors / pairs whose label is negative; miner: keep positives with s < max_neg_i + margin and negatives with
//   s > min_pos_i - margin (empty sets reduce to -FLT_MAX / +FLT_MAX like tf.reduce_max / reduce_min on empty ragged rows);
//   loss_i = log(1 + sum_pos exp(-a (s - l))) / a + log(1 + sum_neg exp(b (s - l))) / b; the Keras loss is the batch mean.
// Backward: w_ij = d(mean loss)/d(s_ij) (mining masks are constants), df_i = sum_j (w_ij + w_ji) f_j.
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

struct MsParams {
    float alpha, beta, lambda, margin;
    int use_miner, ignore_diag, ignore_neg;
};

// one workgroup per anchor i: similarities to every j, mining thresholds, loss_i and w_ij = d(loss_i)/d(s_ij) * grad_scale
__global__ void __launch_bounds__(256) ms_rows_kernel(const float* __restrict__ f, const int32_t* __restrict__ labels, float* __restrict__ loss_rows,
                                                      float* __restrict__ wmat, int B, int D, MsParams p, float grad_scale) {
    extern __shared__ float sim[];     // [B]
    __shared__ float red[4];
    const int i = blockIdx.x;
    const int li = labels[i];
    const float* fi = f + (int64_t)i * D;
    float mx_neg = -FLT_MAX, mn_pos = FLT_MAX;
    for (int j = threadIdx.x; j < B; j += 256) {
        const float* fj = f + (int64_t)j * D;
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += fi[d] * fj[d];
        sim[j] = s;
        const int lj = labels[j];
        const bool valid = !(p.ignore_diag && j == i) && !(p.ignore_neg && li < 0);
        if (valid) {
            if (lj == li) mn_pos = fminf(mn_pos, s);
            else mx_neg = fmaxf(mx_neg, s);
        }
    }
    mx_neg = block_reduce(mx_neg, red, 1);
    mn_pos = block_reduce(mn_pos, red, 2);
    const float pos_thr = mx_neg + p.margin, neg_thr = mn_pos - p.margin;
    float sp = 0.f, sn = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) {
        const float s = sim[j];
        const int lj = labels[j];
        const bool valid = !(p.ignore_diag && j == i) && !(p.ignore_neg && li < 0);
        float e = 0.f;
        if (valid && lj == li && (!p.use_miner || s < pos_thr)) { e = expf(-p.alpha * (s - p.lambda)); sp += e; e = -e; }
        else if (valid && lj != li && (!p.use_miner || s > neg_thr)) { e = expf(p.beta * (s - p.lambda)); sn += e; }
        sim[j] = e;      // signed exponential: negative for a kept positive pair, positive for a kept negative pair
    }
    sp = block_reduce(sp, red, 0);
    sn = block_reduce(sn, red, 0);
    if (threadIdx.x == 0) loss_rows[i] = logf(1.0f + sp) / p.alpha + logf(1.0f + sn) / p.beta;
    if (wmat) {
        const float cp = grad_scale / (1.0f + sp), cn = grad_scale / (1.0f + sn);
        for (int j = threadIdx.x; j < B; j += 256) {
            const float e = sim[j];
            wmat[(int64_t)i * B + j] = e < 0.f ? e * cp : e * cn;     // d/ds of log(1 + sum exp(-a(s-l)))/a = -exp(..)/(1+sum)
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

int chb_multi_similarity_loss(const float* emb, const int32_t* labels, float* loss_rows, float* workspace, float* d_emb, int B, int D,
                              float pos_scale, float neg_scale, float threshold, float miner_margin, int use_miner, int ignore_diag,
                              int ignore_negative_labels, void* stream) {
    if (B < 0 || D <= 0 || pos_scale <= 0.f || neg_scale <= 0.f) return CHB_EINVAL;
    if (B == 0) return CHB_OK;
    if (!emb || !labels || !loss_rows || (d_emb && !workspace)) return CHB_EINVAL;
    if ((size_t)B * sizeof(float) > 60 * 1024) return CHB_EUNSUPPORTED;      // one row of similarities lives in LDS
    MsParams p{pos_scale, neg_scale, threshold, miner_margin, use_miner ? 1 : 0, ignore_diag ? 1 : 0, ignore_negative_labels ? 1 : 0};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ms_rows_kernel, dim3(B), dim3(256), (size_t)B * sizeof(float), s, emb, labels, loss_rows, d_emb ? workspace : nullptr,
                       B, D, p, 1.0f / (float)B);
    if (d_emb) hipLaunchKernelGGL(ms_grad_kernel, dim3(B), dim3(256), 0, s, emb, workspace, d_emb, B, D);
    CHB_LAUNCH_CHECK();
    return CHB_OK;
}

}  // extern "C"
