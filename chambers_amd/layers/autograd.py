"""torch.autograd.Function wrappers over the C ABI: what makes the stand-alone Keras-style layers TRAINABLE (the reference's
EncoderLayer / MultiHeadAttention / Dense / LayerNormalization / Dropout / LearnedEmbedding1D / ConcatEmbedding are
differentiable layers any user model can compose and fit: chambers/layers/transformer.py:8-77, layers/attention.py:27-127,
layers/embedding.py:156-261).

torch supplies the tape (which Function ran on which tensor) and the memory; every forward and every backward below is a call
into libchambers_hip.so - the same GEMM / attention / LayerNorm kernels the whole-model engine (chambers_amd/engine.py) drives,
at the same rounding points: bf16 MFMA operands, fp32 accumulation, fp32 residual stream, bf16 storage of q/k/v, o, LayerNorm
outputs, gelu(a) and gelu'(a), bf16 dY operands in backward.  No Function falls back to torch arithmetic.

Layer variables are leaf tensors with requires_grad (chambers_amd/_keras_like.Variable); after `loss.backward()` the gradients
are in `variable.value.grad`, and `chambers_amd.optimizers.AdamW.apply_gradients` consumes them."""
import ctypes
import weakref

import torch

from .. import _lib
from .. import kernels as K


def _pad64(n):
    return (int(n) + 63) // 64 * 64


def _bf16_rows(x2d, pad=True):
    """bf16 copy of a [M, K] matrix in a buffer of ceil(M / 64) * 64 rows whose tail rows are zero: the weight-gradient GEMM reduces
    over whole 64-row steps, and zero rows add nothing.  Returns (buffer, view of the first M rows).  pad=False (nothing will be
    differentiated): no zero fill, no tail rows."""
    m, k = x2d.shape
    if not pad:
        buf = K.cast_bf16(x2d.contiguous())
        buf = buf.clone() if buf.data_ptr() == x2d.data_ptr() else buf       # never hand the caller's own storage on as "the copy"
        return buf, buf
    buf = torch.zeros((_pad64(m), k), dtype=torch.bfloat16, device=x2d.device)
    K.cast_bf16(x2d, out=buf)
    return buf, buf[:m]


# bf16 operand images of a layer's fp32 kernel, kept while the kernel has not changed: repeated calls of a stand-alone layer
# (inference, or several forwards per optimizer step) re-use them instead of casting and transposing the weights again.  A change is
# seen through torch's version counter of the tensor (any torch write: set_weights, load) or through `weights_written()`, which the
# optimizer calls after its HIP update (raw-pointer writes that torch's counter cannot see).
_OPERANDS = {}
_GENERATION = [0]
_OPERANDS_MAX = 512


def weights_written():
    """Every cached operand image is stale (called by AdamW.apply_gradients after chb_adamw has rewritten the variables)."""
    _GENERATION[0] += 1
    _OPERANDS.clear()


def _weight_operands(w_kn, n_pad):
    """fp32 [K, N] master -> bf16 [N_pad, K] (forward B operand: rows = output features) and bf16 [K, N_pad] (dgrad B operand).
    Pad columns are zero.  Data movement + one rounding; runs inside Function.forward (no tape)."""
    k, n = w_kn.shape
    key = (id(w_kn), n_pad)           # the tensor OBJECT (a variable's leaf tensor), not its address: freed memory is re-used
    hit = _OPERANDS.get(key)
    if hit is not None and hit[4]() is w_kn and hit[0] == w_kn._version and hit[1] == _GENERATION[0] and hit[5] == w_kn.data_ptr():
        return hit[2], hit[3]
    wb = K.cast_bf16(w_kn.detach().contiguous())
    if n_pad != n:
        full = torch.zeros((k, n_pad), dtype=torch.bfloat16, device=w_kn.device)
        K.store_rows(full[:, :n], wb)
        wb = full
    wt = torch.empty((n_pad, k), dtype=torch.bfloat16, device=w_kn.device)
    wt.copy_(wb.t())          # a strided device copy (data movement)
    if len(_OPERANDS) >= _OPERANDS_MAX:
        _OPERANDS.clear()
    _OPERANDS[key] = (w_kn._version, _GENERATION[0], wt, wb, weakref.ref(w_kn), w_kn.data_ptr())
    return wt, wb


class LinearFn(torch.autograd.Function):
    """y = act(x . W + b) with W fp32 [K, N] (the Keras `kernel` layout), through chb_gemm_nt; act in (None, "gelu", "tanh").
    Backward: dz = dy (.) act' (bf16, one rounding), dx = dz . W^T (chb_gemm_nt), dW = x^T . dz (chb_gemm_tn), db = column sums.
    Output fp32 [M, N] unless out_bf16 (the storage dtype of an activation that only feeds another GEMM)."""

    @staticmethod
    def forward(ctx, x, w_kn, bias, act, out_bf16):
        k, n = w_kn.shape
        if x.shape[-1] != k:
            raise ValueError("LinearFn: input width %d, kernel %s" % (x.shape[-1], tuple(w_kn.shape)))
        if k % 64:
            raise ValueError("the MFMA GEMM needs in_features % 64 == 0 (got %d)" % k)
        n_pad = _pad64(n)
        lead = x.shape[:-1]
        taped = any(ctx.needs_input_grad[:3])       # all False under torch.no_grad() or when nothing upstream requires grad
        a_buf, a = _bf16_rows(x.detach().reshape(-1, k), pad=taped)
        m = a.shape[0]
        wt, wkn = _weight_operands(w_kn, n_pad)
        b = None
        if bias is not None:
            b = bias.detach().reshape(-1).contiguous()
            if n_pad != n:
                bp = torch.zeros(n_pad, dtype=torch.float32, device=b.device)
                bp[:n] = b
                b = bp
        store_bf16 = bool(out_bf16) and act != "tanh"
        out = torch.empty((m, n_pad), dtype=torch.bfloat16 if store_bf16 else torch.float32, device=a.device)
        aux = None
        if act == "gelu":
            aux = torch.empty((m, n_pad), dtype=torch.bfloat16, device=a.device)
            K.gemm_nt(a, wt, out, bias=b, epilogue=K.EPI_GELU, aux=aux)
        elif act in (None, "linear", "tanh"):
            K.gemm_nt(a, wt, out, bias=b)
            if act == "tanh":
                K.tanh_fwd(out)
        else:
            raise ValueError("unsupported activation %r" % (act,))
        ctx.act, ctx.n, ctx.n_pad, ctx.k, ctx.lead = act, n, n_pad, k, lead
        ctx.has_bias = bias is not None
        ctx.x_dtype = x.dtype
        ctx.m = m
        if taped:
            ctx.save_for_backward(a_buf, wkn, aux, out if act == "tanh" else None)
        y = out if n_pad == n else out[:, :n]
        return y.reshape(*lead, n)

    @staticmethod
    def backward(ctx, dy):
        a_buf, wkn, aux, ytanh = ctx.saved_tensors
        n, n_pad, k, m = ctx.n, ctx.n_pad, ctx.k, ctx.m
        dy2 = dy.reshape(m, n)
        if n_pad != n:                                   # pad columns carry no gradient (their weights are zeros that stay zeros)
            full = torch.zeros((m, n_pad), dtype=dy2.dtype, device=dy2.device)
            full[:, :n] = dy2
            dy2 = full
        dy2 = dy2.contiguous()
        dz_buf = torch.zeros((_pad64(m), n_pad), dtype=torch.bfloat16, device=dy2.device)      # zero tail rows for the weight-gradient GEMM
        dz = dz_buf[:m]
        if ctx.act == "gelu":
            K.scale_by_bf16(dy2, aux, out=dz_buf)
        elif ctx.act == "tanh":
            K.tanh_bwd(K.cast_f32(dy2), ytanh, dz_buf)
        else:
            K.cast_bf16(dy2, out=dz_buf)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((m, k), dtype=torch.bfloat16 if ctx.x_dtype == torch.bfloat16 else torch.float32, device=dz.device)
            K.gemm_nt(dz, wkn, dx)
            dx = dx.reshape(*ctx.lead, k)
        if ctx.needs_input_grad[1]:
            dwp = torch.zeros((k, n_pad), dtype=torch.float32, device=dz.device)
            K.gemm_tn(a_buf, dz_buf, dwp)
            dw = dwp if n_pad == n else dwp[:, :n].contiguous()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            dbp = torch.zeros(n_pad, dtype=torch.float32, device=dz.device)
            K.colsum(dz, dbp)
            db = dbp[:n] if n_pad != n else dbp
        return dx, dw, db, None, None


class LinearResidualFn(torch.autograd.Function):
    """y = resid + dropout(x . W + b): the block's two projections that end in a residual add (layers/transformer.py:57-58,69,76),
    one GEMM with the CHB_EPI_RESID epilogue.  Backward: d_resid = dy; dz = bf16(dropout-backward of dy) (chb_dropout_bwd_bf16, the
    same element index as the epilogue), then dgrad / wgrad / bias sums as LinearFn."""

    @staticmethod
    def forward(ctx, x, w_kn, bias, resid, rate, key):
        k, n = w_kn.shape
        if k % 64 or n % 64:
            raise ValueError("the fused residual projection needs in / out features % 64 == 0 (got %d -> %d)" % (k, n))
        lead = resid.shape[:-1]
        taped = any(ctx.needs_input_grad[:4])
        a_buf, a = _bf16_rows(x.detach().reshape(-1, k), pad=taped)
        m = a.shape[0]
        wt, wkn = _weight_operands(w_kn, n)
        r = resid.detach().reshape(m, n)
        if r.dtype != torch.float32:
            r = K.cast_f32(r)
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
        K.gemm_nt(a, wt, out, bias=bias.detach().reshape(-1).contiguous(), epilogue=K.EPI_RESID, resid=r.contiguous(), drop_rate=float(rate),
                  drop_key=int(key) if rate else 0)
        ctx.rate, ctx.key, ctx.k, ctx.n, ctx.lead, ctx.x_dtype, ctx.x_lead = float(rate), int(key), k, n, lead, x.dtype, x.shape[:-1]
        ctx.r_dtype, ctx.m = resid.dtype, m
        if taped:
            ctx.save_for_backward(a_buf, wkn)
        return out.reshape(*lead, n)

    @staticmethod
    def backward(ctx, dy):
        a_buf, wkn = ctx.saved_tensors
        m, k, n = ctx.m, ctx.k, ctx.n
        dy2 = K.cast_f32(dy.reshape(m, n)).contiguous()
        dz_buf = torch.zeros((_pad64(m), n), dtype=torch.bfloat16, device=dy2.device)
        dz = dz_buf[:m]
        K.dropout_bwd(dy2, dz_buf, m, n, ctx.rate, ctx.key if ctx.rate else 0)
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((m, k), dtype=torch.bfloat16 if ctx.x_dtype == torch.bfloat16 else torch.float32, device=dz.device)
            K.gemm_nt(dz, wkn, dx)
            dx = dx.reshape(*ctx.x_lead, k)
        if ctx.needs_input_grad[1]:
            dw = torch.zeros((k, n), dtype=torch.float32, device=dz.device)
            K.gemm_tn(a_buf, dz_buf, dw)
        if ctx.needs_input_grad[2]:
            db = torch.zeros(n, dtype=torch.float32, device=dz.device)
            K.colsum(dz, db)
        if ctx.needs_input_grad[3]:
            dres = (K.cast_bf16(dy2) if ctx.r_dtype == torch.bfloat16 else dy2).reshape(*ctx.lead, n)
        return dx, dw, db, dres, None, None


class LayerNormFn(torch.autograd.Function):
    """LayerNormalization over the last axis: fp32 rows in, bf16 rows out (the operand of the GEMM that follows), statistics in
    fp32 (chb_layernorm_fwd / chb_layernorm_bwd)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        d = x.shape[-1]
        x2 = x.detach().reshape(-1, d)
        x2 = (K.cast_f32(x2) if x2.dtype != torch.float32 else x2).contiguous()
        m = x2.shape[0]
        y = torch.empty((m, d), dtype=torch.bfloat16, device=x2.device)
        mean = torch.empty(m, dtype=torch.float32, device=x2.device)
        rstd = torch.empty(m, dtype=torch.float32, device=x2.device)
        g = gamma.detach().contiguous()
        K.layernorm_fwd(x2, d, g, beta.detach().contiguous(), y, mean, rstd, m, d, float(eps))
        ctx.shape, ctx.x_dtype = x.shape, x.dtype
        ctx.save_for_backward(x2, mean, rstd, g)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, g = ctx.saved_tensors
        m, d = x2.shape
        dyb = K.cast_bf16(dy.reshape(m, d)).contiguous()
        dx = torch.empty((m, d), dtype=torch.float32, device=x2.device)
        dgamma = torch.zeros(d, dtype=torch.float32, device=x2.device)
        dbeta = torch.zeros(d, dtype=torch.float32, device=x2.device)
        K.layernorm_bwd(dyb, x2, d, mean, rstd, g, dx, d, False, dgamma, dbeta, m, d)
        if ctx.x_dtype == torch.bfloat16:
            dx = K.cast_bf16(dx)
        return dx.reshape(ctx.shape), dgamma, dbeta, None


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd)) with dropout on the probabilities, times v, on the packed bf16 [B*T, 3*H*hd] projection
    (chb_attention_fwd / chb_attention_bwd; layers/attention.py:7-23,120-122).  Up to 224 tokens the forward also saves the keep
    bits of its dropout mask for the backward."""

    @staticmethod
    def forward(ctx, qkv, b, t, h, hd, rate, key):
        q = qkv.detach()
        q = (K.cast_bf16(q) if q.dtype != torch.bfloat16 else q).contiguous()
        d = h * hd
        o = torch.empty((b * t, d), dtype=torch.bfloat16, device=q.device)
        lse = torch.empty(b * h * t, dtype=torch.float32, device=q.device)
        bits = K.attention_drop_bits(b, t, h, device=q.device) if (rate and t <= 224) else None
        K.attention_fwd(q, o, lse, b, t, h, hd, float(rate), int(key) if rate else 0, drop_bits=bits)
        ctx.dims = (b, t, h, hd, float(rate), int(key) if rate else 0)
        ctx.in_dtype = qkv.dtype
        ctx.save_for_backward(q, o, lse, bits)
        return o

    @staticmethod
    def backward(ctx, do):
        q, o, lse, bits = ctx.saved_tensors
        b, t, h, hd, rate, key = ctx.dims
        dob = K.cast_bf16(do).contiguous()
        dqkv = torch.zeros_like(q)
        K.attention_bwd(q, o, dob, lse, dqkv, b, t, h, hd, rate, key, drop_bits=bits)
        if ctx.in_dtype == torch.float32:
            dqkv = K.cast_f32(dqkv)
        return dqkv, None, None, None, None, None, None


class AttentionGeneralFn(torch.autograd.Function):
    """keras Attention semantics beyond the ViT's use (layers/attention.py:99-153): value / query masks, causal mask, cross-attention
    (Tq != Tk), through chb_attention_general_fwd / _bwd.  q [B*Tq, H*hd], k / v [B*Tk, H*hd]; masks uint8 [B, T] or None."""

    @staticmethod
    def forward(ctx, q, k, v, b, tq, tk, h, hd, vmask, qmask, causal, rate, key, scale=0.0):
        qq, kk, vv = (K.cast_bf16(t.detach()).contiguous() for t in (q, k, v))
        o, lse = K.attention_general_fwd(qq, kk, vv, b, tq, tk, h, hd, vmask, qmask, causal, rate, key if rate else 0, scale=scale)
        ctx.dims = (b, tq, tk, h, hd, bool(causal), float(rate), int(key) if rate else 0)
        ctx.scale = float(scale)
        ctx.dtypes = (q.dtype, k.dtype, v.dtype)
        ctx.masks = (vmask, qmask)
        ctx.save_for_backward(qq, kk, vv, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        qq, kk, vv, o, lse = ctx.saved_tensors
        b, tq, tk, h, hd, causal, rate, key = ctx.dims
        dq, dk, dv = K.attention_general_bwd(qq, kk, vv, o, K.cast_bf16(do).contiguous(), lse, b, tq, tk, h, hd, ctx.masks[0], ctx.masks[1],
                                             causal, rate, key, scale=ctx.scale)
        out = [K.cast_bf16(g) if dt == torch.bfloat16 else g for g, dt in zip((dq, dk, dv), ctx.dtypes)]
        return (out[0], out[1], out[2]) + (None,) * 11


class DropoutFn(torch.autograd.Function):
    """keras Dropout as a layer: y = x * keep / (1 - rate), mask from the counter hash on the flat element index; the backward is
    the same map on dy (chb_dropout_f32)."""

    @staticmethod
    def forward(ctx, x, rate, key):
        ctx.rate, ctx.key, ctx.dtype = float(rate), int(key), x.dtype
        return K.dropout_f32(K.cast_f32(x.detach()), ctx.rate, ctx.key).reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        dx = K.dropout_f32(K.cast_f32(dy), ctx.rate, ctx.key).reshape(dy.shape)
        return (K.cast_bf16(dx) if ctx.dtype == torch.bfloat16 else dx), None, None


class AddFn(torch.autograd.Function):
    """a + b in fp32 (a residual add that is not fused into a GEMM epilogue); the gradient passes to both unchanged."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.dt = (a.dtype, b.dtype)
        return K.add_f32(K.cast_f32(a.detach()).contiguous(), K.cast_f32(b.detach()).contiguous().reshape(a.shape))

    @staticmethod
    def backward(ctx, dy):
        da = K.cast_bf16(dy) if ctx.dt[0] == torch.bfloat16 else dy
        db = K.cast_bf16(dy) if ctx.dt[1] == torch.bfloat16 else dy
        return da, db


class CastF32Fn(torch.autograd.Function):
    """bf16 -> fp32 (a layer boundary that hands fp32 to its caller); the gradient is rounded to bf16 on the way back."""

    @staticmethod
    def forward(ctx, x):
        ctx.dtype = x.dtype
        return K.cast_f32(x.detach())

    @staticmethod
    def backward(ctx, dy):
        return K.cast_bf16(dy) if ctx.dtype == torch.bfloat16 else dy


class GeluFn(torch.autograd.Function):
    """chambers.activations.gelu on a tensor (activations.py:5-56), exact-erf or tanh form, with its derivative."""

    @staticmethod
    def forward(ctx, x, approximate):
        y, d = K.gelu_f32(K.cast_f32(x.detach()).contiguous(), approximate=approximate, want_derivative=True)
        ctx.save_for_backward(d)
        ctx.dtype = x.dtype
        return y.reshape(x.shape) if x.dtype == torch.float32 else K.cast_bf16(y).reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (d,) = ctx.saved_tensors
        dx = K.mul_f32(K.cast_f32(dy).contiguous().reshape(d.shape), d).reshape(dy.shape)
        return (K.cast_bf16(dx) if ctx.dtype == torch.bfloat16 else dx), None


class AddTableFn(torch.autograd.Function):
    """LearnedEmbedding1D: x [B, N, D] + table [N, D] (layers/embedding.py:176-182); d_table = sum over the batch."""

    @staticmethod
    def forward(ctx, x, table):
        ctx.shape = x.shape
        return K.add_rows_f32(K.cast_f32(x.detach()).contiguous(), table.detach().contiguous())

    @staticmethod
    def backward(ctx, dy):
        dy = K.cast_f32(dy).contiguous()
        period = 1
        for s in ctx.shape[1:]:
            period *= int(s)
        dtab = K.sum_rows_f32(dy, rows=int(ctx.shape[0]), cols=period).reshape(ctx.shape[1:]) if ctx.needs_input_grad[1] else None
        return (dy if ctx.needs_input_grad[0] else None), dtab


class ConcatTokensFn(torch.autograd.Function):
    """ConcatEmbedding along the token axis (layers/embedding.py:100-104 as the ViT uses it: axis=1): [emb | x] or [x | emb] by
    strided row copies; backward: dx = the x rows of dy (strided copy), d_emb = sum over the batch of the embedding rows."""

    @staticmethod
    def forward(ctx, x, emb, left):
        xx = x.detach().contiguous()
        e = emb.detach()
        e = (K.cast_bf16(e) if xx.dtype == torch.bfloat16 else e).contiguous()
        b = xx.shape[0]
        ctx.left, ctx.ne, ctx.x_shape = bool(left), int(e.shape[0]), x.shape
        n, d = int(xx.shape[1]), int(xx.shape[2])
        es = xx.element_size()
        out = torch.empty((b, n + ctx.ne, d), dtype=xx.dtype, device=xx.device)
        e_at, x_at = (0, ctx.ne) if left else (n, 0)
        row = (n + ctx.ne) * d * es
        # the embedding rows are read with stride 0 over the batch: strided row copies, no arithmetic
        _lib.call("chb_copy_rows", _lib.ptr(e), 0, ctypes.c_void_p(out.data_ptr() + e_at * d * es), row, b, ctx.ne * d * es, K._s())
        _lib.call("chb_copy_rows", _lib.ptr(xx), n * d * es, ctypes.c_void_p(out.data_ptr() + x_at * d * es), row, b, n * d * es, K._s())
        return out

    @staticmethod
    def backward(ctx, dy):
        b, n, d = (int(s) for s in ctx.x_shape)
        ne = ctx.ne
        dyf = K.cast_f32(dy).contiguous()
        e_at, x_at = (0, ne) if ctx.left else (n, 0)
        dx = demb = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((b, n, d), dtype=torch.float32, device=dyf.device)
            _lib.call("chb_copy_rows", ctypes.c_void_p(dyf.data_ptr() + x_at * d * 4), (n + ne) * d * 4, _lib.ptr(dx), n * d * 4, b, n * d * 4, K._s())
            if dy.dtype == torch.bfloat16:
                dx = K.cast_bf16(dx)
        if ctx.needs_input_grad[1]:
            demb = K.sum_rows_f32(dyf, rows=b, cols=ne * d, row_stride=(n + ne) * d, offset=e_at * d).reshape(ne, d)
        return dx, demb, None


class TakeTokenFn(torch.autograd.Function):
    """x[:, index, :] of a [B, N, D] sequence - the `cls` pooling of the ViT (vision_transformer.py:182-189) as a strided row copy;
    the backward scatters dy into that row of a zero tensor."""

    @staticmethod
    def forward(ctx, x, index):
        xx = x.detach().contiguous()
        b, n, d = (int(s) for s in xx.shape)
        es = xx.element_size()
        out = torch.empty((b, d), dtype=xx.dtype, device=xx.device)
        _lib.call("chb_copy_rows", ctypes.c_void_p(xx.data_ptr() + int(index) * d * es), n * d * es, _lib.ptr(out), d * es, b, d * es, K._s())
        ctx.shape, ctx.index = (b, n, d), int(index)
        return out

    @staticmethod
    def backward(ctx, dy):
        b, n, d = ctx.shape
        dyc = dy.contiguous()
        es = dyc.element_size()
        dx = torch.zeros((b, n, d), dtype=dyc.dtype, device=dyc.device)
        _lib.call("chb_copy_rows", _lib.ptr(dyc), d * es, ctypes.c_void_p(dx.data_ptr() + ctx.index * d * es), n * d * es, b, d * es, K._s())
        return dx, None
