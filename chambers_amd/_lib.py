"""ctypes binding of libchambers_hip.so (the C ABI declared in include/chambers_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call returns an
error code, an exception is raised.  torch is imported first so that the HIP runtime the
library links against (libamdhip64.so.7) is the instance torch already loaded — streams and
device pointers are then shared between torch (allocation / streams / RCCL) and the kernels.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL: shares libamdhip64 with torch)

from . import _build

c_void_p, c_int, c_int64, c_float, c_uint32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint32

CHB_OK, CHB_EINVAL, CHB_ELAUNCH, CHB_EUNSUPPORTED = 0, -1, -2, -3

PW_INVERT, PW_POSTERIZE, PW_SOLARIZE, PW_SOLARIZE_ADD, PW_BRIGHTNESS, PW_CONTRAST, PW_COLOR = range(7)
(AUG_IDENTITY, AUG_AUTOCONTRAST, AUG_EQUALIZE, AUG_INVERT, AUG_POSTERIZE, AUG_SOLARIZE, AUG_SOLARIZE_ADD, AUG_BRIGHTNESS, AUG_CONTRAST,
 AUG_COLOR, AUG_SHARPNESS, AUG_AFFINE, AUG_CUTOUT) = range(13)
NORM_CAFFE, NORM_TF, NORM_TORCH = 0, 1, 2
EPI_NONE, EPI_GELU, EPI_DGELU, EPI_RESID, EPI_PATCH = range(5)
OUT_BF16, OUT_F32 = 0, 1

P = c_void_p
# name -> argument ctypes (all return int except the two info calls)
PROTOTYPES = {
    "chb_aug_pointwise": [P, P, c_int64, c_int, c_float, c_int, c_int, P],
    "chb_aug_affine": [P, P, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P],
    "chb_aug_cutout": [P, P, c_int, c_int, c_int, c_int, P, c_int, c_int, P],
    "chb_aug_autocontrast": [P, P, c_int, c_int, c_int, c_int, P, P],
    "chb_aug_equalize": [P, P, c_int, c_int, c_int, c_int, P, P],
    "chb_aug_sharpness": [P, P, c_int, c_int, c_int, c_int, c_float, P],
    "chb_aug_dispatch": [P, P, c_int, c_int, c_int, P, c_int, P, P],
    "chb_aug_fused": [P, P, c_int, c_int, c_int, c_int, P, P, P, P, c_int, P],
    "chb_aug_fused_workspace_ints": [c_int, c_int, c_int, c_int],
    "chb_aug_fused_items": [P, P, c_int, c_int, c_int, c_int, P, P, c_int, P, c_int, P],
    "chb_aug_items_sort": [P, c_int, c_int, c_int, c_int, P, P],
    "chb_aug_fused_items_sorted": [P, P, c_int, c_int, c_int, c_int, P, P, c_int, P, c_int, P, P, P],
    "chb_normalize_u8": [P, P, c_int64, c_int, c_int, P],
    "chb_normalize_f32": [P, P, c_int64, c_int, c_int, P],
    "chb_normalize_patchify_bf16": [P, P, c_int, c_int, c_int, c_int, c_int, P],
    "chb_patchify_f32_bf16": [P, P, c_int, c_int, c_int, c_int, P],
    "chb_dropout_mask": [P, c_int64, c_float, c_uint32, P],
    "chb_gemm_nt": [P, c_int64, P, c_int64, P, c_int64, c_int, c_int, c_int, P, c_int, c_int, P, c_int64, P, c_int64,
                    c_int, c_float, c_uint32, P, P],
    "chb_gemm_tn": [P, c_int64, P, c_int64, P, c_int64, c_int, c_int, c_int, P],
    "chb_gemm_tn_ws": [P, c_int64, P, c_int64, P, c_int64, c_int, c_int, c_int, P, c_int64, c_int, P, P],
    "chb_gemm_tn_fold": [P, c_int64, P, c_int64, c_int, c_int, c_int, P],
    "chb_gemm_tn_fold_multi": [P, c_int, P],
    "chb_layernorm_fwd": [P, c_int64, P, P, P, P, P, c_int, c_int, c_float, P],
    "chb_layernorm_bwd": [P, P, c_int64, P, P, P, P, c_int64, c_int, P, P, c_int, c_int, P, P, c_float, c_uint32, c_int, P],
    "chb_attention_fwd": [P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint32, P, P],
    "chb_attention_bwd": [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint32, P, P, P, P],
    "chb_cls_row": [P, P, P, c_int, c_int, c_int, c_float, c_uint32, P],
    "chb_token_row": [P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint32, P],
    "chb_embed_bwd": [P, P, P, P, c_int, c_int, c_int, c_float, c_uint32, P],
    "chb_embed_bwd_tokens": [P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint32, P],
    "chb_dropout_bwd_bf16": [P, c_int64, P, c_int, c_int, c_float, c_uint32, P],
    "chb_colsum_bf16": [P, c_int64, P, c_int, c_int, P],
    "chb_softmax_ce": [P, c_int64, P, P, P, c_int64, c_int, c_int, c_float, P],
    "chb_pool_tokens": [P, P, P, c_int, c_int, c_int, c_int, P],
    "chb_l2_normalize_fwd": [P, P, P, c_int, c_int, P],
    "chb_l2_normalize_bwd": [P, P, P, P, c_int, c_int, P],
    "chb_multi_similarity_loss": [P, P, P, P, P, c_int, c_int, c_float, c_float, c_float, c_float, c_int, c_int, c_int, P],
    "chb_multi_similarity_loss_matrix": [P, P, P, P, c_int, c_float, c_float, c_float, c_float, c_int, c_int, P],
    "chb_contrastive_loss": [P, P, P, P, P, c_int, c_int, c_float, c_float, c_float, c_float, c_int, c_int, c_int, P],
    "chb_ntxent_loss": [P, P, P, P, P, c_int, c_int, c_float, c_int, P],
    "chb_resize": [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "chb_resize_ragged": [P, c_int64, P, P, c_int, P, c_int, c_int, c_int, c_int, P],
    "chb_crop_flip": [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, c_int, P, P],
    "chb_rescale": [P, c_int, P, c_int64, c_float, c_float, P],
    "chb_pool_tokens_bwd": [P, P, P, c_int, c_int, c_int, c_int, P],
    "chb_tanh_fwd": [P, P, c_int64, P],
    "chb_tanh_bwd": [P, P, P, c_int64, P],
    "chb_cast_transpose": [P, P, P, P, c_int, c_int, P],
    "chb_adamw": [P, P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_float, c_int, P],
    "chb_zero_f32": [P, c_int64, P],
    "chb_add_f32": [P, P, P, c_int64, P],
    "chb_cast_f32_bf16": [P, P, c_int64, P],
    "chb_cast_bf16_f32": [P, P, c_int64, P],
    "chb_copy_rows": [P, c_int64, P, c_int64, c_int64, c_int64, P],
    "chb_softmax_f32": [P, c_int64, P, c_int64, c_int, c_int, P],
    "chb_set_option": [ctypes.c_char_p, c_int],
    "chb_gemm_tile_queue_reset": [P],
    "chb_attention_general_fwd": [P, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_float, c_uint32,
                                  c_float, P],
    "chb_attention_general_bwd": [P, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_int64, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P,
                                  c_int, c_float, c_uint32, c_float, P],
    "chb_gelu_f32": [P, P, P, c_int64, c_int, P],
    "chb_mul_f32": [P, P, P, c_int64, P],
    "chb_scale_by_bf16": [P, c_int, P, P, c_int64, P],
    "chb_dropout_f32": [P, P, c_int64, c_float, ctypes.c_uint32, P],
    "chb_add_rows_f32": [P, P, P, c_int64, c_int64, P],
    "chb_sum_rows_f32": [P, c_int64, c_int64, c_int64, P, P],
    "chb_axpby_f32": [P, c_float, P, c_float, P, c_int64, P],
    "chb_add_rows_bf16": [P, c_int64, P, c_int64, c_int64, c_int, P],
    "chb_vit_block_fwd": [P, c_int, P],
    "chb_vit_block_bwd": [P, c_int, P, P],
    "chb_side_stream_join": [P, P],
    "chb_profile_enable": [c_int],
    "chb_profile_collect": [P, c_int, P],
}
INFO_SYMBOLS = ["chb_version", "chb_build_arch"]

c_int32 = ctypes.c_int32


class VitBlock(ctypes.Structure):
    """`chb_vit_block` of include/chambers_hip.h, field for field (tests/test_abi_signatures.py compares the two declarations and the
    byte layout a C compiler gives the header's struct)."""
    _fields_ = ([(n, c_int32) for n in ("B", "N", "H", "hd", "D", "FF", "M", "Mg", "Mp")] +
                [("eps", c_float), ("drop_rate", c_float)] +
                [(n, c_uint32) for n in ("key_attn", "key_proj", "key_mlp", "key_prev_mlp")] +
                [("emit_dz", c_int32), ("reserved", c_int32)] +
                [(n, P) for n in ("ln1_gamma", "ln1_beta", "ln2_gamma", "ln2_beta", "qkv_bias", "proj_bias", "fc1_bias", "fc2_bias",
                                  "qkv_wt", "proj_wt", "fc1_wt", "fc2_wt", "qkv_w", "proj_w", "fc1_w", "fc2_w", "x_in", "x_out",
                                  "h1", "qkv", "o", "h2", "a1", "u", "mean1", "rstd1", "lse", "xmid", "mean2", "rstd2", "drop_bits",
                                  "g_ln1_gamma", "g_ln1_beta", "g_ln2_gamma", "g_ln2_beta", "g_qkv_bias", "g_proj_bias", "g_fc1_bias",
                                  "g_prev_fc2_bias", "g_qkv_w", "g_proj_w", "g_fc1_w", "g_fc2_w", "dx", "dz", "da1", "dh", "d_o", "dqkv",
                                  "tn_ws", "tn_ws_side")] +
                [("tn_ws_bytes", c_int64), ("tn_ws4", P)])


class TnFoldItem(ctypes.Structure):
    """`chb_tn_fold_item` of include/chambers_hip.h."""
    _fields_ = [("workspace", P), ("workspace_bytes", c_int64), ("dW", P), ("ldw", c_int64), ("M", c_int32), ("Kd", c_int32), ("Nd", c_int32),
                ("reserved", c_int32)]


class ProfileRecord(ctypes.Structure):
    """`chb_profile_record` of include/chambers_hip.h."""
    _fields_ = [("kind", c_int32), ("family", c_int32), ("epilogue", c_int32), ("out_dtype", c_int32), ("m", c_int64), ("n", c_int64),
                ("k", c_int64), ("ms", c_float), ("start_ms", c_float)]


def profile_enable(on):
    call("chb_profile_enable", 1 if on else 0)


def profile_collect():
    """Records of the current profiling period as a list of dicts (waits for the recorded events)."""
    n = ctypes.c_int(0)
    call("chb_profile_collect", None, 0, ctypes.byref(n))
    recs = (ProfileRecord * max(n.value, 1))()
    call("chb_profile_collect", ctypes.cast(recs, c_void_p), n.value, ctypes.byref(n))
    return [{f: getattr(recs[i], f) for f, _t in ProfileRecord._fields_} for i in range(n.value)]

_lib = None


class ChambersHipError(RuntimeError):
    pass


def lib_path():
    return _build.LIB_PATH


def load():
    """Load (once) and type the shared library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ChambersHipError(
            "libchambers_hip.so not found at %s — run `python -m chambers_amd._build` "
            "(or __graft_entry__.build()); there is no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.chb_version.restype = c_int
    lib.chb_aug_fused_workspace_ints.restype = c_int64       # a size, not a status
    lib.chb_build_arch.restype = ctypes.c_char_p
    _lib = lib
    return lib


_ERR = {CHB_EINVAL: "invalid argument", CHB_ELAUNCH: "kernel launch failed", CHB_EUNSUPPORTED: "unsupported shape"}


def check(name, code):
    """Map C-ABI return codes onto the reference's Python error conventions."""
    if code == CHB_OK:
        return
    msg = "%s: %s (code %d)" % (name, _ERR.get(code, "error"), code)
    if code in (CHB_EINVAL, CHB_EUNSUPPORTED):
        raise ValueError(msg)
    raise ChambersHipError(msg)


def stream_ptr():
    """Current torch HIP stream as a raw hipStream_t."""
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise ChambersHipError(
                "chambers_amd kernels run on an MI355X only; got a %s tensor (no CPU fallback)" % t.device)


def set_option(name, value):
    """A/B switch of the library (see chb_set_option in include/chambers_hip.h); name without the CHB_ prefix."""
    call("chb_set_option", name.encode(), int(value))


def aug_fused_workspace_ints(b, h, w, n_tables):
    """int32 elements of chb_aug_fused's workspace (host arithmetic inside the library: it owns the slice count)."""
    return int(load().chb_aug_fused_workspace_ints(int(b), int(h), int(w), int(n_tables)))


def call(name, *args):
    lib = load()
    check(name, getattr(lib, name)(*args))
