"""Thin tensor-level wrappers over the C ABI (runtime.py).  Allocation and stream selection are
torch's; every computation happens in libsitk.so's HIP kernels."""
import ctypes as C

import torch

from . import runtime as rt
from .runtime import BF16, F16, F32, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_DGELU, EPI_STORE  # noqa: F401


def pad8(n):
    return (n + 7) // 8 * 8


def pad64(n):
    return (n + 63) // 64 * 64


def _rowmap(rm):
    return rt.RowMap(*rm) if rm else rt.RowMap(0, 0, 0)


# ---- gather / layout -------------------------------------------------------------------------------
def gather_tokens(x_bvc, table_pv, dtype, ld=None, mean=None, std=None):
    """(B, 40962, 4) fp32 + (P, V) uint16 table -> (B*P, ld) tokens of `dtype` (zero padded).
    mean/std: optional (C,) fp32 device tensors -> (x - mean) / std fused in front of the gather."""
    if mean is not None:
        rt.require_cuda(x_bvc, table_pv, mean, std)
        B, nv, Cc = x_bvc.shape
        P, V = table_pv.shape
        code = rt.dtype_code(dtype)
        ld = ld or pad64(V * Cc)
        out = torch.empty((B * P, ld), dtype=rt.torch_dtype(code), device=x_bvc.device)
        rt.check(rt.lib.sitk_gather_tokens_norm(x_bvc.data_ptr(), table_pv.data_ptr(), mean.data_ptr(), std.data_ptr(),
                                                out.data_ptr(), B, nv, Cc, P, V, ld, code, rt.stream_ptr()))
        return out
    rt.require_cuda(x_bvc, table_pv)
    B, nv, Cc = x_bvc.shape
    P, V = table_pv.shape
    assert x_bvc.dtype == torch.float32 and x_bvc.is_contiguous()
    assert table_pv.dtype in (torch.uint16, torch.int16) and table_pv.is_contiguous()  # int16 = same bits
    code = rt.dtype_code(dtype)
    ld = ld or pad64(V * Cc)
    out = torch.empty((B * P, ld), dtype=rt.torch_dtype(code), device=x_bvc.device)
    rt.check(rt.lib.sitk_gather_tokens(x_bvc.data_ptr(), table_pv.data_ptr(), out.data_ptr(), B, nv, Cc, P, V, ld,
                                       code, rt.stream_ptr()))
    return out


def patchify(x_bcpv, dtype, ld=None):
    rt.require_cuda(x_bcpv)
    B, Cc, P, V = x_bcpv.shape
    assert x_bcpv.dtype == torch.float32
    x_bcpv = x_bcpv.contiguous()
    code = rt.dtype_code(dtype)
    ld = ld or pad64(V * Cc)
    out = torch.empty((B * P, ld), dtype=rt.torch_dtype(code), device=x_bcpv.device)
    rt.check(rt.lib.sitk_patchify(x_bcpv.data_ptr(), out.data_ptr(), B, Cc, P, V, ld, code, rt.stream_ptr()))
    return out


def cast_rows(src, dtype, ld=None):
    rt.require_cuda(src)
    assert src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1
    rows, cols = src.shape
    code = rt.dtype_code(dtype)
    ld = ld or pad8(cols)
    out = torch.empty((rows, ld), dtype=rt.torch_dtype(code), device=src.device)
    rt.check(rt.lib.sitk_cast_rows(src.data_ptr(), src.stride(0), out.data_ptr(), ld, rows, cols, code, rt.stream_ptr()))
    return out


def stage_weight(w, dtype, ldc=None, want_c=True, want_t=True):
    rt.require_cuda(w)
    assert w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 2
    rows, cols = w.shape
    code = rt.dtype_code(dtype)
    td = rt.torch_dtype(code)
    ldc = ldc or pad8(cols)
    ldt = pad8(rows)
    wc = torch.empty((rows, ldc), dtype=td, device=w.device) if want_c else None
    wt = torch.empty((cols, ldt), dtype=td, device=w.device) if want_t else None
    rt.check(rt.lib.sitk_stage_weight(w.data_ptr(), rows, cols, rt.ptr(wc), ldc, rt.ptr(wt), ldt, code, rt.stream_ptr()))
    return wc, wt


# ---- GEMMs -----------------------------------------------------------------------------------------
def gemm_nt(A, W, out, dtype, M=None, N=None, K=None, epilogue=EPI_STORE, bias=None, aux=None, out2=None,
            amap=None, omap=None, auxmap=None):
    """out[m, n] = sum_k A[m, k] W[n, k] (+ epilogue).  A: compute dtype or fp32; W: compute dtype;
    out: compute dtype or fp32.  2-D tensors with unit inner stride; leading dims from strides.
    EPI_BIAS_GELU (ABI 9): `out` receives gelu'(u) - 1/2 of u = acc + bias (the CENTRED derivative EPI_DGELU's `aux` expects),
    NOT the pre-activation u; `out2` receives gelu(u)."""
    rt.require_cuda(A, W, out, bias, aux, out2)
    code = rt.dtype_code(dtype)
    d = rt.GemmDesc()
    d.M = M if M is not None else A.shape[0]
    d.N = N if N is not None else W.shape[0]
    d.K = K if K is not None else W.shape[1]
    d.A, d.lda, d.a_is_f32, d.amap = A.data_ptr(), A.stride(0), int(A.dtype == torch.float32), _rowmap(amap)
    d.W, d.ldw = W.data_ptr(), W.stride(0)
    d.epilogue = epilogue
    d.out, d.ldo, d.out_is_f32, d.omap = out.data_ptr(), out.stride(0), int(out.dtype == torch.float32), _rowmap(omap)
    d.out2 = rt.ptr(out2)
    d.bias = rt.ptr(bias)
    d.aux, d.ldaux, d.auxmap = rt.ptr(aux), (aux.stride(0) if aux is not None else 0), _rowmap(auxmap)
    rt.check(rt.lib.sitk_gemm_nt(C.byref(d), code, rt.stream_ptr()))
    return out


def gemm_wgrad(dY, X, dW, dtype, db=None, M=None, N=None, K=None, dymap=None, xmap=None):
    """dW[n, k] += sum_m dY[m, n] X[m, k]; db[n] += sum_m dY[m, n]."""
    rt.require_cuda(dY, X, dW, db)
    code = rt.dtype_code(dtype)
    d = rt.WgradDesc()
    d.M = M if M is not None else dY.shape[0]
    d.N = N if N is not None else dW.shape[0]
    d.K = K if K is not None else dW.shape[1]
    d.dY, d.lddy, d.dy_is_f32, d.dymap = dY.data_ptr(), dY.stride(0), int(dY.dtype == torch.float32), _rowmap(dymap)
    d.X, d.ldx, d.xmap = X.data_ptr(), X.stride(0), _rowmap(xmap)
    d.dW, d.lddw, d.db = dW.data_ptr(), dW.stride(0), rt.ptr(db)
    assert dW.dtype == torch.float32
    rt.check(rt.lib.sitk_gemm_wgrad(C.byref(d), code, rt.stream_ptr()))
    return dW


def gemm_wgrad_group(problems, dtype, workspace=None):
    """problems: list of dicts(dY, X, dW, db=None[, dymap=(group, stride, offset), M, K]): all weight gradients in ONE launch.
    workspace: None (64x64 tiles + atomics), a uint8 tensor, or "auto" (allocate what the large-tile
    slab path asks for)."""
    code = rt.dtype_code(dtype)
    arr = (rt.WgradDesc * len(problems))()
    for d, p in zip(arr, problems):
        dY, X, dW, db = p["dY"], p["X"], p["dW"], p.get("db")
        rt.require_cuda(dY, X, dW, db)
        d.M, d.N, d.K = p.get("M", dY.shape[0]), dW.shape[0], p.get("K", dW.shape[1])
        d.dY, d.lddy, d.dy_is_f32, d.dymap = dY.data_ptr(), dY.stride(0), int(dY.dtype == torch.float32), _rowmap(p.get("dymap"))
        d.X, d.ldx, d.xmap = X.data_ptr(), X.stride(0), _rowmap(None)
        d.dW, d.lddw, d.db = dW.data_ptr(), dW.stride(0), rt.ptr(db)
    if workspace == "auto":
        nbytes = rt.lib.sitk_gemm_wgrad_group_ws_bytes(arr, len(problems), code)
        workspace = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=problems[0]["dW"].device)
    if workspace is None:
        rt.check(rt.lib.sitk_gemm_wgrad_group(arr, len(problems), code, rt.stream_ptr()))
    else:
        rt.check(rt.lib.sitk_gemm_wgrad_group_ws(arr, len(problems), code, workspace.data_ptr(), workspace.numel(),
                                                 rt.stream_ptr()))


# ---- LayerNorm -------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, dtype):
    rt.require_cuda(x, gamma, beta)
    rows, D = x.shape
    code = rt.dtype_code(dtype)
    y = torch.empty((rows, D), dtype=rt.torch_dtype(code), device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    rt.check(rt.lib.sitk_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                       rstd.data_ptr(), rows, D, code, rt.stream_ptr()))
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dres, dgamma, dbeta, dtype, dx=None, dx_c=None, partials=None):
    rows, D = x.shape
    code = rt.dtype_code(dtype)
    if dx is None:
        dx = torch.empty_like(x)
    if partials is not None:
        assert partials.numel() >= rt.lib.sitk_layernorm_bwd_partial_floats(rows, D)
    rt.check(rt.lib.sitk_layernorm_bwd(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                       rt.ptr(dres), dx.data_ptr(), rt.ptr(dx_c), dgamma.data_ptr(), dbeta.data_ptr(),
                                       rt.ptr(partials), rows, D, code, rt.stream_ptr()))
    return dx


def layernorm_bwd_partial_floats(rows, D):
    return rt.lib.sitk_layernorm_bwd_partial_floats(rows, D)


# ---- fused LayerNorm + MLP (+ residual) ------------------------------------------------------------
def mlp_fused_supported(D, M, dtype):
    return bool(rt.lib.sitk_mlp_fused_supported(D, M, rt.dtype_code(dtype)))


def mlp_fwd(x, ln_w, ln_b, w1_c, b1, w2_c, b2, dtype, save=True, want_g=False):
    """out = x + gelu(LN(x) W1^T + b1) W2^T + b2; returns (out, h, mean, rstd, gd, g) (saved tensors or None):
    gd = gelu'(u) - 1/2 (CENTRED: backward multiplies by gd + 1/2), g = gelu(u) of the pre-activation u = LN(x) W1^T + b1
    (ABI 9: the derivative is saved, not u -- a caller that reads `gd` as u gets wrong values)."""
    rt.require_cuda(x, ln_w, ln_b, w1_c, b1, w2_c, b2)
    rows, D = x.shape
    M = w1_c.shape[0]
    code = rt.dtype_code(dtype)
    td = rt.torch_dtype(code)
    out = torch.empty_like(x)
    h = torch.empty((rows, D), dtype=td, device=x.device) if save else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save else None
    gd = torch.empty((rows, M), dtype=td, device=x.device) if save else None      # gelu'(u) - 1/2, not u (ABI 9)
    g = torch.empty((rows, M), dtype=td, device=x.device) if want_g else None
    rt.check(rt.lib.sitk_mlp_fwd(x.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), w1_c.data_ptr(), b1.data_ptr(),
                                 w2_c.data_ptr(), b2.data_ptr(), rt.ptr(h), rt.ptr(mean), rt.ptr(rstd), rt.ptr(gd), rt.ptr(g),
                                 out.data_ptr(), rows, D, M, code, rt.stream_ptr()))
    return out, h, mean, rstd, gd, g


def attn_out_mlp_fused_supported(rows, D, I, M, dtype):
    return bool(rt.lib.sitk_attn_out_mlp_fused_supported(rows, D, I, M, rt.dtype_code(dtype)))


def attn_out_mlp_fwd(o_c, wo_c, bo, x, ln_w, ln_b, w1_c, b1, w2_c, b2, dtype, save=True, want_g=False):
    """x_mid = x + o Wo^T + bo; out = x_mid + gelu(LN(x_mid) W1^T + b1) W2^T + b2.
    Returns (out, xmid, h, mean, rstd, gd, g) (gd = gelu'(u) - 1/2, see mlp_fwd)."""
    rt.require_cuda(o_c, wo_c, bo, x, ln_w, ln_b, w1_c, b1, w2_c, b2)
    rows, D = x.shape
    M, I = w1_c.shape[0], o_c.shape[1]
    code = rt.dtype_code(dtype)
    td = rt.torch_dtype(code)
    out, xmid = torch.empty_like(x), torch.empty_like(x)
    h = torch.empty((rows, D), dtype=td, device=x.device) if save else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save else None
    gd = torch.empty((rows, M), dtype=td, device=x.device) if save else None      # gelu'(u) - 1/2, not u (ABI 9)
    g = torch.empty((rows, M), dtype=td, device=x.device) if want_g else None
    rt.check(rt.lib.sitk_attn_out_mlp_fwd(o_c.data_ptr(), wo_c.data_ptr(), bo.data_ptr(), x.data_ptr(), xmid.data_ptr(),
                                          ln_w.data_ptr(), ln_b.data_ptr(), w1_c.data_ptr(), b1.data_ptr(), w2_c.data_ptr(),
                                          b2.data_ptr(), rt.ptr(h), rt.ptr(mean), rt.ptr(rstd), rt.ptr(gd), rt.ptr(g),
                                          out.data_ptr(), rows, D, I, M, code, rt.stream_ptr()))
    return out, xmid, h, mean, rstd, gd, g


def attn_out_mlp_next_fwd(o_c, wo_c, bo, x, ln_w, ln_b, w1_c, b1, w2_c, b2, n_ln_w, n_ln_b, n_wqkv_c, dtype, want_g=False):
    """attn_out_mlp_fwd + the next block's LayerNorm and to_qkv.  Returns (out, xmid, h, mean, rstd, gd, g, n_h, n_mean,
    n_rstd, n_qkv)."""
    rows, D = x.shape
    M, I, N3 = w1_c.shape[0], o_c.shape[1], n_wqkv_c.shape[0]
    code = rt.dtype_code(dtype)
    td = rt.torch_dtype(code)
    dev = x.device
    out, xmid = torch.empty_like(x), torch.empty_like(x)
    h, n_h = torch.empty((rows, D), dtype=td, device=dev), torch.empty((rows, D), dtype=td, device=dev)
    mean, rstd, n_mean, n_rstd = (torch.empty(rows, dtype=torch.float32, device=dev) for _ in range(4))
    gd = torch.empty((rows, M), dtype=td, device=dev)      # gelu'(u) - 1/2 (ABI 9), not the pre-activation
    g = torch.empty((rows, M), dtype=td, device=dev) if want_g else None
    n_qkv = torch.empty((rows, N3), dtype=td, device=dev)
    rt.check(rt.lib.sitk_attn_out_mlp_next_fwd(
        o_c.data_ptr(), wo_c.data_ptr(), bo.data_ptr(), x.data_ptr(), xmid.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(),
        w1_c.data_ptr(), b1.data_ptr(), w2_c.data_ptr(), b2.data_ptr(), h.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
        gd.data_ptr(), rt.ptr(g), out.data_ptr(), n_ln_w.data_ptr(), n_ln_b.data_ptr(), n_wqkv_c.data_ptr(), n_h.data_ptr(),
        n_mean.data_ptr(), n_rstd.data_ptr(), n_qkv.data_ptr(), N3, rows, D, I, M, code, rt.stream_ptr()))
    return out, xmid, h, mean, rstd, gd, g, n_h, n_mean, n_rstd, n_qkv


def mlp_bwd(dy, dy_c, x, mean, rstd, ln_w, w2t_c, w1t_c, gd, dtype):
    """gd = the gelu'(u) - 1/2 saved by the fused forward.  Returns (dx, dx_c, du, partials (workgroups, 2, D)).
    dy_c = None: the kernel rounds dy itself and also returns the compute-dtype copy it wrote (sitk_mlp_bwd_cast):
    (dx, dx_c, du, partials, dy_c)."""
    rows, D = x.shape
    if dy_c is None:
        return _mlp_bwd_cast(dy, x, mean, rstd, ln_w, w2t_c, w1t_c, gd, dtype)
    M = gd.shape[1]
    code = rt.dtype_code(dtype)
    du = torch.empty_like(gd)
    dx = torch.empty_like(x)
    dx_c = torch.empty((rows, D), dtype=gd.dtype, device=x.device)
    nfl = rt.lib.sitk_mlp_bwd_partial_floats(rows)
    partials = torch.empty(nfl, dtype=torch.float32, device=x.device)
    rt.check(rt.lib.sitk_mlp_bwd(dy.data_ptr(), dy_c.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                 ln_w.data_ptr(), w2t_c.data_ptr(), w1t_c.data_ptr(), gd.data_ptr(), du.data_ptr(),
                                 dx.data_ptr(), dx_c.data_ptr(), partials.data_ptr(), rows, D, M, code, rt.stream_ptr()))
    return dx, dx_c, du, partials.view(-1, 2, D)


def _mlp_bwd_cast(dy, x, mean, rstd, ln_w, w2t_c, w1t_c, gd, dtype):
    rows, D = x.shape
    M = gd.shape[1]
    du = torch.empty_like(gd)
    dx = torch.empty_like(x)
    dx_c = torch.empty((rows, D), dtype=gd.dtype, device=x.device)
    dy_c = torch.empty((rows, D), dtype=gd.dtype, device=x.device)
    partials = torch.empty(rt.lib.sitk_mlp_bwd_partial_floats(rows), dtype=torch.float32, device=x.device)
    rt.check(rt.lib.sitk_mlp_bwd_cast(dy.data_ptr(), dy_c.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                      ln_w.data_ptr(), w2t_c.data_ptr(), w1t_c.data_ptr(), gd.data_ptr(), du.data_ptr(),
                                      dx.data_ptr(), dx_c.data_ptr(), partials.data_ptr(), rows, D, M, rt.dtype_code(dtype),
                                      rt.stream_ptr()))
    return dx, dx_c, du, partials.view(-1, 2, D), dy_c


# ---- fused LayerNorm + to_qkv ----------------------------------------------------------------------
def ln_gemm_fused_supported(D, N, dtype):
    return bool(rt.lib.sitk_ln_gemm_fused_supported(D, N, rt.dtype_code(dtype)))


def ln_gemm_fwd(x, ln_w, ln_b, w_c, dtype, save=True):
    """y = LN(x) W^T; returns (y, h, mean, rstd)"""
    rt.require_cuda(x, ln_w, ln_b, w_c)
    rows, D = x.shape
    N = w_c.shape[0]
    code = rt.dtype_code(dtype)
    td = rt.torch_dtype(code)
    y = torch.empty((rows, N), dtype=td, device=x.device)
    h = torch.empty((rows, D), dtype=td, device=x.device) if save else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save else None
    rt.check(rt.lib.sitk_ln_gemm_fwd(x.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), w_c.data_ptr(), rt.ptr(h), rt.ptr(mean),
                                     rt.ptr(rstd), y.data_ptr(), rows, D, N, code, rt.stream_ptr()))
    return y, h, mean, rstd


def ln_gemm_bwd(dy, wt_c, x, mean, rstd, ln_w, dres, dtype):
    """dx = dres + LN'(dy W); returns (dx, dx_c, partials (workgroups, 2, D))"""
    rows, D = x.shape
    N = dy.shape[1]
    code = rt.dtype_code(dtype)
    dx = torch.empty_like(x)
    dx_c = torch.empty((rows, D), dtype=dy.dtype, device=x.device)
    partials = torch.empty(rt.lib.sitk_ln_gemm_bwd_partial_floats(rows), dtype=torch.float32, device=x.device)
    rt.check(rt.lib.sitk_ln_gemm_bwd(dy.data_ptr(), wt_c.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                     ln_w.data_ptr(), rt.ptr(dres), dx.data_ptr(), dx_c.data_ptr(), partials.data_ptr(), rows,
                                     D, N, code, rt.stream_ptr()))
    return dx, dx_c, partials.view(-1, 2, D)


def ln_gemm_mlp_bwd_supported(rows, D, N, M, dtype):
    return bool(rt.lib.sitk_ln_gemm_mlp_bwd_supported(rows, D, N, M, rt.dtype_code(dtype)))


def ln_gemm_mlp_bwd(dqkv, wqkv_t_c, x, mean1, rstd1, ln1_w, dres, xmid, mean2, rstd2, ln2_w, w2t_c, w1t_c, gd, dtype):
    """ln_gemm_bwd (layer l) + mlp_bwd (layer l - 1, on the dx / dx_c the first half writes) in ONE launch.
    Returns (dx, dx_c, partials1, dx_mid, dx_mid_c, du, partials2) -- the outputs of the two calls, bit for bit."""
    rows, D = x.shape
    N, M = dqkv.shape[1], gd.shape[1]
    code = rt.dtype_code(dtype)
    dx, dx_mid = torch.empty_like(x), torch.empty_like(x)
    dx_c = torch.empty((rows, D), dtype=dqkv.dtype, device=x.device)
    dx_mid_c = torch.empty_like(dx_c)
    du = torch.empty_like(gd)
    p1 = torch.empty(rt.lib.sitk_ln_gemm_bwd_partial_floats(rows), dtype=torch.float32, device=x.device)
    p2 = torch.empty(rt.lib.sitk_mlp_bwd_partial_floats(rows), dtype=torch.float32, device=x.device)
    rt.check(rt.lib.sitk_ln_gemm_mlp_bwd(dqkv.data_ptr(), wqkv_t_c.data_ptr(), x.data_ptr(), mean1.data_ptr(), rstd1.data_ptr(),
                                         ln1_w.data_ptr(), rt.ptr(dres), dx.data_ptr(), dx_c.data_ptr(), p1.data_ptr(), N,
                                         xmid.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(), ln2_w.data_ptr(), w2t_c.data_ptr(),
                                         w1t_c.data_ptr(), gd.data_ptr(), du.data_ptr(), dx_mid.data_ptr(), dx_mid_c.data_ptr(),
                                         p2.data_ptr(), rows, D, M, code, rt.stream_ptr()))
    return dx, dx_c, p1.view(-1, 2, D), dx_mid, dx_mid_c, du, p2.view(-1, 2, D)


# ---- attention -------------------------------------------------------------------------------------
def attention_fwd(qkv, B, N, H, scale, dtype):
    code = rt.dtype_code(dtype)
    o = torch.empty((B * N, H * 64), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    rt.check(rt.lib.sitk_attention_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, H, scale, code, rt.stream_ptr()))
    return o, lse


def attention_bwd(qkv, o, d_o, lse, B, N, H, scale, dtype):
    code = rt.dtype_code(dtype)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    rt.check(rt.lib.sitk_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                       dqkv.data_ptr(), B, N, H, scale, code, rt.stream_ptr()))
    return dqkv


def attention_bwd_proj_supported(N, D, dtype):
    return bool(rt.lib.sitk_attention_bwd_proj_supported(N, D, rt.dtype_code(dtype)))


def attention_bwd_proj(qkv, o, dxmid, wo_t, lse, B, N, H, scale, dtype):
    """attention backward with d_o = dxmid @ Wo formed inside the query-side kernel; returns (dqkv, d_o)."""
    code = rt.dtype_code(dtype)
    D = dxmid.shape[1]
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    d_o = torch.empty_like(o)
    rt.check(rt.lib.sitk_attention_bwd_proj(qkv.data_ptr(), o.data_ptr(), dxmid.data_ptr(), wo_t.data_ptr(), d_o.data_ptr(),
                                            lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), B, N, H, D, scale, code,
                                            rt.stream_ptr()))
    return dqkv, d_o


# ---- encoder ---------------------------------------------------------------------------------------
def encoder_cfg(B, N, dim, depth, heads, mlp_dim, dtype):
    return rt.EncoderCfg(B, N, dim, depth, heads, mlp_dim, rt.dtype_code(dtype))


def layer_param_array(per_layer_tensors):
    """per_layer_tensors: list (depth) of sequences of 11 fp32 tensors in rt.LAYER_FIELDS order."""
    arr = (rt.LayerParams * len(per_layer_tensors))()
    for i, ts in enumerate(per_layer_tensors):
        assert len(ts) == len(rt.LAYER_FIELDS)
        for k, t in zip(rt.LAYER_FIELDS, ts):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda, k
            setattr(arr[i], k, t.data_ptr())
    return arr


def encoder_workspace(cfg, device):
    ab = rt.lib.sitk_encoder_acts_bytes(C.byref(cfg))
    sb = rt.lib.sitk_encoder_scratch_bytes(C.byref(cfg))
    if ab == 0 or sb == 0:
        raise rt.SitkError(f"encoder workspace query failed: {rt.lib.sitk_last_error().decode()}")
    return (torch.empty(ab, dtype=torch.uint8, device=device), torch.empty(sb, dtype=torch.uint8, device=device))


def encoder_fwd(cfg, params, x_in, x_out, acts, scratch, save=True):
    rt.check(rt.lib.sitk_encoder_fwd(C.byref(cfg), params, x_in.data_ptr(), x_out.data_ptr(), acts.data_ptr(),
                                     acts.numel(), scratch.data_ptr(), scratch.numel(), int(save), rt.stream_ptr()))
    return x_out


def encoder_bwd(cfg, params, grads, x_in, dx, acts, scratch, layer_begin=0, layer_end=None):
    layer_end = cfg.depth if layer_end is None else layer_end
    rt.check(rt.lib.sitk_encoder_bwd(C.byref(cfg), params, grads, x_in.data_ptr(), dx.data_ptr(), acts.data_ptr(),
                                     acts.numel(), scratch.data_ptr(), scratch.numel(), layer_begin, layer_end,
                                     rt.stream_ptr()))
    return dx


def wgrad_desc(dY, X, dW, db=None, M=None, N=None, K=None, dymap=None):
    d = rt.WgradDesc()
    d.M = M if M is not None else dY.shape[0]
    d.N = N if N is not None else dW.shape[0]
    d.K = K if K is not None else dW.shape[1]
    d.dY, d.lddy, d.dy_is_f32, d.dymap = dY.data_ptr(), dY.stride(0), int(dY.dtype == torch.float32), _rowmap(dymap)
    d.X, d.ldx, d.xmap = X.data_ptr(), X.stride(0), _rowmap(None)
    d.dW, d.lddw, d.db = dW.data_ptr(), dW.stride(0), rt.ptr(db)
    return d


def encoder_bwd_embed(cfg, params, grads, x_in, dx, acts, scratch, layer_begin, layer_end, tokens, dW, db, dx_c, P, extra=None,
                      overlap=None):
    """encoder_bwd with the patch embedding's weight gradient (dW (D, ld) fp32 += d(x_in)[rows 1..P]^T tokens, db) taken
    into the slice's one weight-gradient launch; `extra`: more WgradDesc problems for the same launch.  Returns
    (embed taken, extra taken): what was not taken the caller runs itself (gemm_wgrad).  overlap: handle of
    sitk_overlap_create -> the first finished layers' weight gradients run on its side stream beside the chain."""
    d = rt.WgradDesc()
    d.M, d.N, d.K = tokens.shape[0], dW.shape[0], dW.shape[1]
    d.dY, d.lddy, d.dy_is_f32, d.dymap = 0, dW.shape[0], 0, _rowmap((P, P + 1, 1))
    d.X, d.ldx, d.xmap = tokens.data_ptr(), tokens.stride(0), _rowmap(None)
    d.dW, d.lddw, d.db = dW.data_ptr(), dW.stride(0), rt.ptr(db)
    done, xdone = C.c_int(0), C.c_int(0)
    extra = extra or []
    xarr = (rt.WgradDesc * max(1, len(extra)))(*extra)
    rt.check(rt.lib.sitk_encoder_bwd_overlap(C.byref(cfg), params, grads, x_in.data_ptr(), dx.data_ptr(), acts.data_ptr(),
                                             acts.numel(), scratch.data_ptr(), scratch.numel(), layer_begin, layer_end,
                                             C.byref(d), dx_c.data_ptr(), C.byref(done), xarr, len(extra), C.byref(xdone),
                                             overlap, rt.stream_ptr()))
    return bool(done.value), bool(xdone.value)


def embed_cls_rows(x, cls_token, pos, B, N, D):
    rt.check(rt.lib.sitk_embed_cls_rows(x.data_ptr(), cls_token.data_ptr(), pos.data_ptr(), B, N, D, rt.stream_ptr()))
    return x


# ---- head / loss -----------------------------------------------------------------------------------
def head_fwd(x, ln_w, ln_b, w, b, B, N, D, pool_mean):
    ncls = w.shape[0]
    logits = torch.empty((B, ncls), dtype=torch.float32, device=x.device)
    rt.check(rt.lib.sitk_head_fwd(x.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), w.data_ptr(), b.data_ptr(),
                                  logits.data_ptr(), B, N, D, ncls, int(pool_mean), rt.stream_ptr()))
    return logits


def head_bwd(x, ln_w, ln_b, w, dlogits, dx, d_ln_w, d_ln_b, d_w, d_b, B, N, D, pool_mean):
    ncls = w.shape[0]
    ws = torch.empty(rt.lib.sitk_head_ws_floats(B, D, ncls), dtype=torch.float32, device=x.device)   # ordered sums, no atomics
    rt.check(rt.lib.sitk_head_bwd(x.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), w.data_ptr(), dlogits.data_ptr(),
                                  dx.data_ptr(), d_ln_w.data_ptr(), d_ln_b.data_ptr(), d_w.data_ptr(), d_b.data_ptr(),
                                  B, N, D, ncls, int(pool_mean), ws.data_ptr(), rt.stream_ptr()))
    return dx


def head_loss_fwd_bwd(x, ln_w, ln_b, w, b, target, loss, dx, d_ln_w, d_ln_b, d_w, d_b, B, N, D, pool_mean, l1=False,
                      ordered=True, grad_scale=None):
    """pool + head + loss and their backward in one launch; returns logits (B, n_classes).  ordered: per-sample terms of
    the parameter gradients / loss go through a workspace and are added in sample order (bitwise reproducible).
    grad_scale: (2,) fp32 device tensor -> every gradient comes out multiplied by the power of two S the call picks from
    the batch and leaves there as {S, 1 / S} (loss scaling of the f16 compute mode)."""
    ncls = w.shape[0]
    logits = torch.empty((B, ncls), dtype=torch.float32, device=x.device)
    ws = torch.empty(rt.lib.sitk_head_ws_floats(B, D, ncls), dtype=torch.float32, device=x.device) if ordered else None
    rt.check(rt.lib.sitk_head_loss_fwd_bwd(x.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), w.data_ptr(), b.data_ptr(),
                                           target.data_ptr(), logits.data_ptr(), loss.data_ptr(), dx.data_ptr(),
                                           d_ln_w.data_ptr(), d_ln_b.data_ptr(), d_w.data_ptr(), d_b.data_ptr(), B, N, D, ncls,
                                           int(pool_mean), int(l1), rt.ptr(ws), rt.ptr(grad_scale), rt.stream_ptr()))
    return logits


def head_loss_fwd_bwd_deferred(x, ln_w, ln_b, w, b, target, dx, B, N, D, pool_mean, l1=False, grad_scale=None):
    """head_loss_fwd_bwd without the sum of the per-sample gradient terms: returns (logits, ws); head_finalize(ws, ...) adds
    them (and the loss) later, on any stream ordered behind this call."""
    ncls = w.shape[0]
    logits = torch.empty((B, ncls), dtype=torch.float32, device=x.device)
    ws = torch.empty(rt.lib.sitk_head_ws_floats(B, D, ncls), dtype=torch.float32, device=x.device)
    rt.check(rt.lib.sitk_head_loss_fwd_bwd_deferred(x.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), w.data_ptr(), b.data_ptr(),
                                                    target.data_ptr(), logits.data_ptr(), dx.data_ptr(), B, N, D, ncls,
                                                    int(pool_mean), int(l1), ws.data_ptr(), rt.ptr(grad_scale), rt.stream_ptr()))
    return logits, ws


def head_finalize(ws, B, D, ncls, d_ln_w, d_ln_b, d_w, d_b, loss):
    rt.check(rt.lib.sitk_head_finalize(ws.data_ptr(), B, D, ncls, d_ln_w.data_ptr(), d_ln_b.data_ptr(), d_w.data_ptr(),
                                       d_b.data_ptr(), loss.data_ptr(), rt.stream_ptr()))


def loss_fwd_bwd(pred, target, loss, dpred, l1=False):
    rt.check(rt.lib.sitk_loss_fwd_bwd(pred.data_ptr(), target.data_ptr(), loss.data_ptr(), dpred.data_ptr(),
                                      pred.numel(), int(l1), rt.stream_ptr()))


def colsum_f32(x2d, out):
    rows, cols = x2d.shape
    rt.check(rt.lib.sitk_colsum_f32(x2d.data_ptr(), rows, cols, x2d.stride(0), out.data_ptr(), rt.stream_ptr()))
    return out


def masked_colsum(x2d, flag_a, flag_b, out, dtype, cols=None):
    rows = x2d.shape[0]
    cols = cols or x2d.shape[1]
    rt.check(rt.lib.sitk_masked_colsum(x2d.data_ptr(), x2d.stride(0), int(x2d.dtype == torch.float32),
                                       rt.dtype_code(dtype), flag_a.data_ptr(), rt.ptr(flag_b), rows, cols,
                                       out.data_ptr(), rt.stream_ptr()))
    return out


# ---- MPP -------------------------------------------------------------------------------------------
def mpp_corrupt(tokens, masked, swap_draw, random_patches, replace_draw, mask_token, B, P, K, dtype, ld=None):
    code = rt.dtype_code(dtype)
    ld = ld or pad64(K)
    out = torch.empty((B * P, ld), dtype=rt.torch_dtype(code), device=tokens.device)
    rt.check(rt.lib.sitk_mpp_corrupt(tokens.data_ptr(), masked.data_ptr(), rt.ptr(swap_draw), rt.ptr(random_patches),
                                     replace_draw.data_ptr(), mask_token.data_ptr(), out.data_ptr(), B, P, K, ld, code,
                                     rt.stream_ptr()))
    return out


def mpp_loss_fwd_bwd(out, tokens, masked, loss, dout, n_masked_total):
    rows, K = out.shape
    rt.check(rt.lib.sitk_mpp_loss_fwd_bwd(out.data_ptr(), tokens.data_ptr(), masked.data_ptr(), loss.data_ptr(),
                                          dout.data_ptr(), rows, K, n_masked_total, rt.stream_ptr()))


# ---- optimizers ------------------------------------------------------------------------------------
def sgd_step(param, grad, buf, lr, momentum=0.0, weight_decay=0.0, nesterov=False, grad_scale=1.0):
    rt.check(rt.lib.sitk_sgd_step(param.data_ptr(), grad.data_ptr(), rt.ptr(buf), param.numel(), lr, momentum,
                                  weight_decay, int(nesterov), grad_scale, rt.stream_ptr()))


def adam_step(param, grad, m, v, lr, beta1, beta2, eps, weight_decay, decoupled, step, grad_scale=1.0):
    rt.check(rt.lib.sitk_adam_step(param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), param.numel(), lr,
                                   beta1, beta2, eps, weight_decay, int(decoupled), step, grad_scale, rt.stream_ptr()))
