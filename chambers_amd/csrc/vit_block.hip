// One encoder block of the Vision Transformer per C-ABI call, and the launch profiler.
//
// chb_vit_block_fwd / chb_vit_block_bwd issue exactly the launches the Python engine issued one by one through ctypes (the same
// chb_* entries, same arguments, same order, same stream), so their results are bit-identical to the call-by-call step; what changes
// is the host side: one Python -> ctypes transition per block and direction instead of 7 forward / 13 backward (a ViT-B/16 train step
// was ~500 transitions and 11 - 45 ms of host time, depending on the box, against 69 ms of GPU time).
//
// Replaces, per call, the Keras op sequence of EncoderLayer.call, pre-norm branch (layers/transformer.py:53-77: norm1 -> MultiHeadAttention
// (layers/attention.py:113-125) -> dropout -> residual; norm2 -> dense1 (gelu) -> dense2 -> dropout -> residual) and its gradient.
//
// The profiler (chb_profile_enable / chb_profile_collect) brackets every GEMM launch with HIP events ON THE LAUNCH STREAM from inside
// chb_gemm_nt / chb_gemm_tn_ws: bench.py's live roofline no longer needs a Python wrapper around each GEMM.
#include "common.hpp"
#include "../../include/chambers_hip.h"
#include <atomic>
#include <mutex>
#include <vector>

// ------------------------------------------------------------------------------------------------ profiler
namespace {

struct ProfRec {
    int kind, family, epi, out_dtype;
    int64_t m, n, k;
    hipEvent_t start, stop;
};

std::atomic<int> g_prof_on{0};
std::mutex g_prof_mutex;
std::vector<ProfRec> g_prof_recs;          // records of the current period
std::vector<hipEvent_t> g_prof_pool;       // events created so far (re-used across periods)
size_t g_prof_pool_used = 0;
constexpr size_t PROF_MAX_RECORDS = 1 << 16;

hipEvent_t prof_event() {                  // g_prof_mutex held
    if (g_prof_pool_used < g_prof_pool.size()) return g_prof_pool[g_prof_pool_used++];
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_prof_pool.push_back(e);
    ++g_prof_pool_used;
    return e;
}

}  // namespace

int chb_prof_begin(int kind, int epi, int out_dtype, int64_t m, int64_t n, int64_t k, hipStream_t s) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return -1;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (!g_prof_on.load(std::memory_order_relaxed) || g_prof_recs.size() >= PROF_MAX_RECORDS) return -1;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return -1;     // timing events do not belong in a graph
    ProfRec r;
    r.kind = kind; r.family = 0; r.epi = epi; r.out_dtype = out_dtype; r.m = m; r.n = n; r.k = k;
    r.start = prof_event();
    r.stop = prof_event();
    if (!r.start || !r.stop) return -1;
    if (hipEventRecord(r.start, s) != hipSuccess) return -1;
    g_prof_recs.push_back(r);
    return (int)g_prof_recs.size() - 1;
}

void chb_prof_end(int slot, int family, hipStream_t s) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if ((size_t)slot >= g_prof_recs.size()) return;
    g_prof_recs[slot].family = family;
    (void)hipEventRecord(g_prof_recs[slot].stop, s);
}

extern "C" {

int chb_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (on) {
        g_prof_recs.clear();
        g_prof_pool_used = 0;
    }
    g_prof_on.store(on ? 1 : 0, std::memory_order_relaxed);
    return CHB_OK;
}

int chb_profile_collect(chb_profile_record* out_host, int max_records, int* n_records_host) {
    if (!n_records_host || max_records < 0 || (max_records > 0 && !out_host)) return CHB_EINVAL;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    const int n = (int)g_prof_recs.size();
    *n_records_host = n;
    for (int r = 0; r < n && r < max_records; ++r) {
        const ProfRec& p = g_prof_recs[r];
        float ms = 0.f, since_first = 0.f;
        if (hipEventSynchronize(p.stop) != hipSuccess || hipEventElapsedTime(&ms, p.start, p.stop) != hipSuccess ||
            hipEventElapsedTime(&since_first, g_prof_recs[0].start, p.start) != hipSuccess)
            return CHB_ELAUNCH;
        out_host[r].kind = p.kind; out_host[r].family = p.family; out_host[r].epilogue = p.epi; out_host[r].out_dtype = p.out_dtype;
        out_host[r].m = p.m; out_host[r].n = p.n; out_host[r].k = p.k;
        out_host[r].ms = ms;
        out_host[r].start_ms = since_first;
    }
    return CHB_OK;
}

// ------------------------------------------------------------------------------------------------ encoder block
#define CHB_TRY(call)                \
    do {                             \
        const int rc__ = (call);     \
        if (rc__ != CHB_OK) return rc__; \
    } while (0)

static int block_args_ok(const chb_vit_block* b) {
    if (!b || b->B <= 0 || b->N <= 0 || b->H <= 0 || b->D <= 0 || b->FF <= 0) return 0;
    if (b->hd != 64 || b->D != b->H * b->hd) return 0;
    if (b->M != b->B * b->N || b->Mg < b->M || b->Mp < b->Mg) return 0;
    if (b->drop_rate < 0.f || b->drop_rate >= 1.f) return 0;
    return 1;
}

int chb_vit_block_fwd(const chb_vit_block* b, int training, void* stream) {
    if (!block_args_ok(b)) return CHB_EINVAL;
    if (!b->x_in || !b->x_out || !b->h1 || !b->qkv || !b->o || !b->lse || !b->xmid || !b->h2 || !b->a1 || !b->u || !b->mean1 || !b->rstd1 ||
        !b->mean2 || !b->rstd2)
        return CHB_EINVAL;
    const int D = b->D, FF = b->FF, M = b->M, Mg = b->Mg;
    const float rate = training ? b->drop_rate : 0.f;
    // x_mid = x_in + dropout(proj(attention(LN1(x_in))))
    CHB_TRY(chb_layernorm_fwd(b->x_in, D, b->ln1_gamma, b->ln1_beta, b->h1, b->mean1, b->rstd1, M, D, b->eps, stream));
    CHB_TRY(chb_gemm_nt(b->h1, D, b->qkv_wt, D, b->qkv, 3 * D, Mg, 3 * D, D, b->qkv_bias, CHB_EPI_NONE, CHB_OUT_BF16, nullptr, 0, nullptr, 0, 0, 0.f, 0u,
                        nullptr, stream));
    CHB_TRY(chb_attention_fwd(b->qkv, b->o, b->lse, b->B, b->N, b->H, b->hd, rate, b->key_attn, rate > 0.f ? b->drop_bits : nullptr, stream));
    CHB_TRY(chb_gemm_nt(b->o, D, b->proj_wt, D, b->xmid, D, Mg, D, D, b->proj_bias, CHB_EPI_RESID, CHB_OUT_F32, nullptr, 0, b->x_in, D, 0, rate,
                        b->key_proj, nullptr, stream));
    // x_out = x_mid + dropout(fc2(gelu(fc1(LN2(x_mid)))))
    CHB_TRY(chb_layernorm_fwd(b->xmid, D, b->ln2_gamma, b->ln2_beta, b->h2, b->mean2, b->rstd2, M, D, b->eps, stream));
    CHB_TRY(chb_gemm_nt(b->h2, D, b->fc1_wt, D, b->u, FF, Mg, FF, D, b->fc1_bias, CHB_EPI_GELU, CHB_OUT_BF16, b->a1, FF, nullptr, 0, 0, 0.f, 0u, nullptr,
                        stream));
    CHB_TRY(chb_gemm_nt(b->u, FF, b->fc2_wt, FF, b->x_out, D, Mg, D, FF, b->fc2_bias, CHB_EPI_RESID, CHB_OUT_F32, nullptr, 0, b->xmid, D, 0, rate,
                        b->key_mlp, nullptr, stream));
    return CHB_OK;
}

}  // extern "C"

// Events of the side stream (weight gradients beside the dgrad chain): created once per process, timing off.
namespace {
struct SideEvents {
    hipEvent_t ready[3] = {nullptr, nullptr, nullptr};   // operand final on the main stream: dz, da1, dqkv
    hipEvent_t read[3] = {nullptr, nullptr, nullptr};    // the side stream's last read of that operand
    bool read_pending[3] = {false, false, false};
    hipEvent_t join = nullptr;
    bool ok = false;
};
SideEvents& side_events() {
    static SideEvents ev;
    static std::once_flag once;
    std::call_once(once, [] {
        bool ok = true;
        for (int i = 0; i < 3; ++i)
            ok = ok && hipEventCreateWithFlags(&ev.ready[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev.read[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&ev.join, hipEventDisableTiming) == hipSuccess;
        ev.ok = ok;
    });
    return ev;
}
}  // namespace

extern "C" {

int chb_vit_block_bwd(const chb_vit_block* b, int phases, void* stream, void* side_stream) {
    if (!block_args_ok(b) || !(phases & 3)) return CHB_EINVAL;
    if (!b->dx || !b->dz || !b->da1 || !b->dh || !b->d_o || !b->dqkv || !b->x_in) return CHB_EINVAL;
    if (side_stream && (!b->tn_ws_side || side_stream == stream)) return CHB_EINVAL;
    const int D = b->D, FF = b->FF, M = b->M, Mg = b->Mg, Mp = b->Mp;
    const float rate = b->drop_rate;
    hipStream_t main_s = (hipStream_t)stream, side_s = (hipStream_t)side_stream;
    SideEvents* ev = nullptr;
    if (side_s) {
        ev = &side_events();
        if (!ev->ok) return CHB_ELAUNCH;
    }
    // weight gradient dW += X^T dY (+ column sums of dY): on the side stream when there is one - it depends only on saved activations
    // and on the operand `which` (0 dz, 1 da1, 2 dqkv) the main stream has just finished, and only the optimizer reads its output
    // with tn_ws4 every weight gradient of the call leaves its split-K planes in a scratch of its own and ONE launch folds them at the
    // end of the call (fold_pending); otherwise each GEMM is followed by its own fold launch
    chb_tn_fold_item pending[4];
    int n_pending = 0;
    auto wgrad = [&](const void* X, int64_t ldx, const void* dY, int64_t ldy, float* dW, int64_t ldw, int Kd, int Nd, float* colsum, int which) -> int {
        float* ws = side_s ? b->tn_ws_side : b->tn_ws;
        int fold = 1;
        if (b->tn_ws4 && n_pending < 4) {
            ws = b->tn_ws4 + (size_t)n_pending * (size_t)(b->tn_ws_bytes / 4);
            fold = 0;
            pending[n_pending].workspace = ws; pending[n_pending].workspace_bytes = b->tn_ws_bytes; pending[n_pending].dW = dW; pending[n_pending].ldw = ldw;
            pending[n_pending].M = Mp; pending[n_pending].Kd = Kd; pending[n_pending].Nd = Nd; pending[n_pending].reserved = 0;
            ++n_pending;
        }
        if (!side_s) return chb_gemm_tn_ws(X, ldx, dY, ldy, dW, ldw, Mp, Kd, Nd, ws, b->tn_ws_bytes, fold, colsum, stream);
        if (hipEventRecord(ev->ready[which], main_s) != hipSuccess || hipStreamWaitEvent(side_s, ev->ready[which], 0) != hipSuccess) return CHB_ELAUNCH;
        const int rc = chb_gemm_tn_ws(X, ldx, dY, ldy, dW, ldw, Mp, Kd, Nd, ws, b->tn_ws_bytes, fold, colsum, side_stream);
        if (rc != CHB_OK) return rc;
        if (hipEventRecord(ev->read[which], side_s) != hipSuccess) return CHB_ELAUNCH;
        ev->read_pending[which] = true;
        return CHB_OK;
    };
    // the main stream is about to overwrite operand `which`: the side stream's read of its previous contents comes first
    auto before_write = [&](int which) -> int {
        if (side_s && ev->read_pending[which]) {
            if (hipStreamWaitEvent(main_s, ev->read[which], 0) != hipSuccess) return CHB_ELAUNCH;
            ev->read_pending[which] = false;
        }
        return CHB_OK;
    };
    if (phases & 1) {
        // ---- MLP branch: b->dz holds the dropout-backward of dx at this block's MLP site (written by the block above / the final norm)
        CHB_TRY(wgrad(b->u, FF, b->dz, D, b->g_fc2_w, D, FF, D, nullptr, 0));
        CHB_TRY(before_write(1));
        CHB_TRY(chb_gemm_nt(b->dz, D, b->fc2_w, D, b->da1, FF, Mg, FF, D, nullptr, CHB_EPI_DGELU, CHB_OUT_BF16, b->a1, FF, nullptr, 0, 0, 0.f, 0u, b->g_fc1_bias,
                            stream));
        CHB_TRY(wgrad(b->h2, D, b->da1, FF, b->g_fc1_w, FF, D, FF, nullptr, 1));
        CHB_TRY(chb_gemm_nt(b->da1, FF, b->fc1_w, FF, b->dh, D, Mg, D, FF, nullptr, CHB_EPI_NONE, CHB_OUT_BF16, nullptr, 0, nullptr, 0, 0, 0.f, 0u, nullptr, stream));
        CHB_TRY(before_write(0));
        CHB_TRY(chb_layernorm_bwd(b->dh, b->xmid, D, b->mean2, b->rstd2, b->ln2_gamma, b->dx, D, 1, b->g_ln2_gamma, b->g_ln2_beta, M, D, b->dz, b->g_proj_bias, rate,
                                  b->key_proj, 0, stream));
        // ---- attention branch, first half: b->dz is now the dropout-backward of dx at the projection site
        CHB_TRY(wgrad(b->o, D, b->dz, D, b->g_proj_w, D, D, D, nullptr, 0));
        CHB_TRY(chb_gemm_nt(b->dz, D, b->proj_w, D, b->d_o, D, Mg, D, D, nullptr, CHB_EPI_NONE, CHB_OUT_BF16, nullptr, 0, nullptr, 0, 0, 0.f, 0u, nullptr, stream));
    }
    if (phases & 2) {
        CHB_TRY(before_write(2));
        CHB_TRY(chb_attention_bwd(b->qkv, b->o, b->d_o, b->lse, b->dqkv, b->B, b->N, b->H, b->hd, rate, b->key_attn, nullptr, nullptr,
                                  rate > 0.f ? b->drop_bits : nullptr, stream));
        CHB_TRY(wgrad(b->h1, D, b->dqkv, 3 * D, b->g_qkv_w, 3 * D, D, 3 * D, b->g_qkv_bias, 2));
        CHB_TRY(chb_gemm_nt(b->dqkv, 3 * D, b->qkv_w, 3 * D, b->dh, D, Mg, D, 3 * D, nullptr, CHB_EPI_NONE, CHB_OUT_BF16, nullptr, 0, nullptr, 0, 0, 0.f, 0u, nullptr, stream));
        if (b->emit_dz) {
            CHB_TRY(before_write(0));
            CHB_TRY(chb_layernorm_bwd(b->dh, b->x_in, D, b->mean1, b->rstd1, b->ln1_gamma, b->dx, D, 1, b->g_ln1_gamma, b->g_ln1_beta, M, D, b->dz, b->g_prev_fc2_bias,
                                      rate, b->key_prev_mlp, 0, stream));
        } else {
            CHB_TRY(chb_layernorm_bwd(b->dh, b->x_in, D, b->mean1, b->rstd1, b->ln1_gamma, b->dx, D, 1, b->g_ln1_gamma, b->g_ln1_beta, M, D, nullptr, nullptr, 0.f, 0u, 0,
                                      stream));
        }
    }
    if (n_pending) CHB_TRY(chb_gemm_tn_fold_multi(pending, n_pending, side_s ? side_stream : stream));
    return CHB_OK;
}

int chb_side_stream_join(void* stream, void* side_stream) {
    if (!side_stream || side_stream == stream) return CHB_EINVAL;
    SideEvents& ev = side_events();
    if (!ev.ok) return CHB_ELAUNCH;
    if (hipEventRecord(ev.join, (hipStream_t)side_stream) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, ev.join, 0) != hipSuccess) return CHB_ELAUNCH;
    for (int i = 0; i < 3; ++i) ev.read_pending[i] = false;      // the main stream is now behind every read the side stream made
    return CHB_OK;
}

}  // extern "C"
