"""ctypes binding of libsitk.so (include/sitk.h) -- the only way the package reaches the GPU.

There is NO fallback: if the shared library is missing or a symbol cannot be resolved the import of
this module raises, and every wrapper raises `SitkError` on a non-zero return code.
Pointers come from `tensor.data_ptr()`, the stream from `torch.cuda.current_stream().cuda_stream`.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SITK_LIB: A/B timing of two builds of the same ABI (tools/kbench.py); the default is the in-tree build
LIB_PATH = os.environ.get("SITK_LIB") or os.path.join(_HERE, "libsitk.so")

F32, BF16, F16 = 0, 1, 2
EPI_STORE, EPI_BIAS_RES, EPI_BIAS_GELU, EPI_DGELU = 0, 1, 2, 3
ABI_VERSION = 12


class SitkError(RuntimeError):
    pass


class RowMap(C.Structure):
    _fields_ = [("group", C.c_int), ("stride", C.c_int), ("offset", C.c_int)]


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("A", C.c_void_p), ("lda", C.c_int), ("a_is_f32", C.c_int), ("amap", RowMap),
                ("W", C.c_void_p), ("ldw", C.c_int), ("epilogue", C.c_int),
                ("out", C.c_void_p), ("ldo", C.c_int), ("out_is_f32", C.c_int), ("omap", RowMap),
                ("out2", C.c_void_p), ("bias", C.c_void_p),
                ("aux", C.c_void_p), ("ldaux", C.c_int), ("auxmap", RowMap)]


class WgradDesc(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("dY", C.c_void_p), ("lddy", C.c_int), ("dy_is_f32", C.c_int), ("dymap", RowMap),
                ("X", C.c_void_p), ("ldx", C.c_int), ("xmap", RowMap),
                ("dW", C.c_void_p), ("lddw", C.c_int), ("db", C.c_void_p)]


class EncoderCfg(C.Structure):
    _fields_ = [("B", C.c_int), ("N", C.c_int), ("dim", C.c_int), ("depth", C.c_int), ("heads", C.c_int),
                ("mlp_dim", C.c_int), ("dtype", C.c_int), ("timeline", C.c_void_p)]


LAYER_FIELDS = ("ln1_w", "ln1_b", "wqkv", "wo", "bo", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2")


class LayerParams(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in LAYER_FIELDS]


_P, _I, _L, _F, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_SIGS = {
    "sitk_abi_version": (C.c_int, []),
    "sitk_last_error": (C.c_char_p, []),
    "sitk_dtype_size": (C.c_int, [_I]),
    "sitk_timeline_create": (C.c_void_p, [_I]),
    "sitk_timeline_destroy": (None, [_P]),
    "sitk_timeline_reset": (None, [_P]),
    "sitk_timeline_mark": (C.c_int, [_P, C.c_char_p, _P]),
    "sitk_timeline_read": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_char_p), _I]),
    "sitk_gather_tokens": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sitk_gather_tokens_norm": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sitk_gather_tokens_idx": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sitk_patchify": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sitk_cast_rows": (C.c_int, [_P, _I, _P, _I, _L, _I, _I, _P]),
    "sitk_stage_weight": (C.c_int, [_P, _I, _I, _P, _I, _P, _I, _I, _P]),
    "sitk_gemm_nt": (C.c_int, [C.POINTER(GemmDesc), _I, _P]),
    "sitk_gemm_wgrad": (C.c_int, [C.POINTER(WgradDesc), _I, _P]),
    "sitk_gemm_wgrad_group": (C.c_int, [C.POINTER(WgradDesc), _I, _I, _P]),
    "sitk_gemm_wgrad_group_ws_bytes": (_Z, [C.POINTER(WgradDesc), _I, _I]),
    "sitk_gemm_wgrad_group_ws": (C.c_int, [C.POINTER(WgradDesc), _I, _I, _P, _Z, _P]),
    "sitk_gemm_wgrad_group_ws_cus": (C.c_int, [C.POINTER(WgradDesc), _I, _I, _P, _Z, _I, _P]),
    "sitk_layernorm_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "sitk_layernorm_bwd_partial_floats": (_Z, [_L, _I]),
    "sitk_layernorm_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "sitk_mlp_fused_supported": (C.c_int, [_I, _I, _I]),
    "sitk_mlp_fwd": (C.c_int, [_P] * 13 + [_L, _I, _I, _I, _P]),
    "sitk_attn_out_mlp_fused_supported": (C.c_int, [_L, _I, _I, _I, _I]),
    "sitk_attn_out_mlp_fwd": (C.c_int, [_P] * 17 + [_L, _I, _I, _I, _I, _P]),
    "sitk_attn_out_mlp_next_fwd": (C.c_int, [_P] * 24 + [_I, _L, _I, _I, _I, _I, _P]),
    "sitk_mlp_bwd_partial_floats": (_Z, [_L]),
    "sitk_mlp_bwd": (C.c_int, [_P] * 13 + [_L, _I, _I, _I, _P]),
    "sitk_mlp_bwd_cast": (C.c_int, [_P] * 13 + [_L, _I, _I, _I, _P]),
    "sitk_ln_gemm_mlp_bwd_supported": (C.c_int, [_L, _I, _I, _I, _I]),
    "sitk_ln_gemm_mlp_bwd": (C.c_int, [_P] * 10 + [_I] + [_P] * 11 + [_L, _I, _I, _I, _P]),
    "sitk_ln_gemm_fused_supported": (C.c_int, [_I, _I, _I]),
    "sitk_ln_gemm_fwd": (C.c_int, [_P] * 8 + [_L, _I, _I, _I, _P]),
    "sitk_ln_gemm_bwd_partial_floats": (_Z, [_L]),
    "sitk_ln_gemm_bwd": (C.c_int, [_P] * 10 + [_L, _I, _I, _I, _P]),
    "sitk_attention_fwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "sitk_attention_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "sitk_attention_bwd_proj_supported": (C.c_int, [_I, _I, _I]),
    "sitk_attention_bwd_proj": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "sitk_attention_bwd_phases": (C.c_int, [_P] * 9 + [_I, _I, _I, _I, _F, _I, _I, _P]),
    "sitk_encoder_acts_bytes": (_Z, [C.POINTER(EncoderCfg)]),
    "sitk_encoder_scratch_bytes": (_Z, [C.POINTER(EncoderCfg)]),
    "sitk_encoder_wgrad_slab_bytes": (_Z, [C.POINTER(EncoderCfg)]),
    "sitk_encoder_fwd": (C.c_int, [C.POINTER(EncoderCfg), C.POINTER(LayerParams), _P, _P, _P, _Z, _P, _Z, _I, _P]),
    "sitk_encoder_bwd": (C.c_int, [C.POINTER(EncoderCfg), C.POINTER(LayerParams), C.POINTER(LayerParams), _P, _P,
                                   _P, _Z, _P, _Z, _I, _I, _P]),
    "sitk_encoder_bwd_embed": (C.c_int, [C.POINTER(EncoderCfg), C.POINTER(LayerParams), C.POINTER(LayerParams), _P, _P,
                                         _P, _Z, _P, _Z, _I, _I, C.POINTER(WgradDesc), _P, C.POINTER(C.c_int), _P]),
    "sitk_encoder_bwd_extra": (C.c_int, [C.POINTER(EncoderCfg), C.POINTER(LayerParams), C.POINTER(LayerParams), _P, _P,
                                         _P, _Z, _P, _Z, _I, _I, C.POINTER(WgradDesc), _P, C.POINTER(C.c_int),
                                         C.POINTER(WgradDesc), _I, C.POINTER(C.c_int), _P]),
    "sitk_overlap_create": (C.c_void_p, [_I, _I, _I]),
    "sitk_overlap_stream": (C.c_void_p, [_P]),
    "sitk_overlap_set_layers": (C.c_int, [_P, _I]),
    "sitk_overlap_fork": (C.c_int, [_P, _P]),
    "sitk_overlap_join": (C.c_int, [_P, _P]),
    "sitk_overlap_set_group": (C.c_int, [_P, _I]),
    "sitk_overlap_side_launches": (C.c_int, [_P]),
    "sitk_overlap_wait_side_launch": (C.c_int, [_P, _I, _P]),
    "sitk_overlap_set_tail_cus": (C.c_int, [_P, _I]),
    "sitk_stream_probe": (C.c_int, [_P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sitk_encoder_stage_weights": (C.c_int, [C.POINTER(EncoderCfg), C.POINTER(LayerParams), _P, _Z, _P]),
    "sitk_overlap_destroy": (None, [_P]),
    "sitk_encoder_bwd_overlap": (C.c_int, [C.POINTER(EncoderCfg), C.POINTER(LayerParams), C.POINTER(LayerParams), _P, _P,
                                           _P, _Z, _P, _Z, _I, _I, C.POINTER(WgradDesc), _P, C.POINTER(C.c_int),
                                           C.POINTER(WgradDesc), _I, C.POINTER(C.c_int), _P, _P]),
    "sitk_embed_cls_rows": (C.c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "sitk_head_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sitk_head_ws_floats": (_Z, [_I, _I, _I]),
    "sitk_head_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "sitk_head_loss_fwd_bwd": (C.c_int, [_P] * 13 + [_I] * 6 + [_P, _P, _P]),
    "sitk_head_loss_fwd_bwd_deferred": (C.c_int, [_P] * 8 + [_I] * 6 + [_P, _P, _P]),
    "sitk_head_finalize": (C.c_int, [_P, _I, _I, _I] + [_P] * 6),
    "sitk_loss_fwd_bwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _P]),
    "sitk_colsum_f32": (C.c_int, [_P, _L, _I, _I, _P, _P]),
    "sitk_colsum_f32_dup": (C.c_int, [_P, _L, _I, _I, _P, _P, _I, _P]),
    "sitk_mpp_corrupt": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sitk_mpp_draw": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _P]),
    "sitk_mpp_gather_corrupt": (C.c_int, [_P] * 13 + [_I, _I, _I, _I, _I, _I, _I, _P]),
    "sitk_mpp_loss_fwd_bwd": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _L, _P]),
    "sitk_mpp_loss_fwd_bwd_ld": (C.c_int, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _L, _I, _L, _F, _P]),
    "sitk_masked_colsum": (C.c_int, [_P, _I, _I, _I, _P, _P, _L, _I, _P, _P]),
    "sitk_dropout_fwd": (C.c_int, [_P, _P, _P, _P, _L, _F, _P, _P]),
    "sitk_dropout_bwd": (C.c_int, [_P, _P, _P, _L, _F, _P]),
    "sitk_gelu_fwd": (C.c_int, [_P, _P, _L, _P]),
    "sitk_gelu_bwd": (C.c_int, [_P, _P, _P, _L, _P]),
    "sitk_sgd_step": (C.c_int, [_P, _P, _P, _L, _F, _F, _F, _I, _F, _P]),
    "sitk_adam_step": (C.c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _I, _F, _P]),
    "sitk_sgd_step_dev": (C.c_int, [_P, _P, _P, _L, _P, _F, _F, _I, _F, _I, _L, _L, _P, _P, _P, _P, _P, _F, _P]),
    "sitk_adam_step_dev": (C.c_int, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _I, _F, _I, _L, _L, _P, _P, _P, _P, _P, _F, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)


def _load():
    if not os.path.exists(LIB_PATH):
        raise SitkError(
            f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()' or "
            f"`make -C {os.path.join(_HERE, 'csrc')}`). There is no CPU/eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype, fn.argtypes = res, args
    if lib.sitk_abi_version() != ABI_VERSION:
        raise SitkError(f"libsitk ABI {lib.sitk_abi_version()} != expected {ABI_VERSION}; rebuild")
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise SitkError(f"libsitk error {rc}: {lib.sitk_last_error().decode()}")


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dtype):
    if isinstance(dtype, int) and not isinstance(dtype, bool) and dtype in (F32, BF16, F16):
        return dtype
    if dtype in ("bf16", torch.bfloat16):
        return BF16
    if dtype in ("f16", "fp16", "half", torch.float16):
        return F16
    if dtype in ("f32", "fp32", torch.float32):
        return F32
    raise ValueError(f"unsupported compute dtype {dtype!r} (use 'bf16', 'f16' or 'f32')")


def torch_dtype(code):
    return {BF16: torch.bfloat16, F16: torch.float16, F32: torch.float32}[code]


def ptr(t):
    return 0 if t is None else t.data_ptr()


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SitkError("sitk: tensors must live on a ROCm/HIP device (no CPU path in the product code)")
