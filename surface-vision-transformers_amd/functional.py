"""torch.autograd bridges over the C ABI: each Function is one fused stage of the hot path
(embedding, encoder, head, generic Linear), so `loss.backward()` of tools/train.py:290 /
tools/pretrain.py:318 runs the HIP backward kernels.  No Function has a CPU branch."""
import math

import torch

from . import ops
from . import runtime as rt


def _zeros_like_all(ts):
    return [torch.zeros_like(t) for t in ts]


def _f16_scale(dy, dtype):
    """Loss scaling of the f16 compute mode on the autograd (drop-in module) path: the incoming gradient is multiplied by
    S = 2^k, k such that max |dy| * S lies in [64, 128), before it becomes a 16-bit MFMA operand, and every result is
    divided by S again (exact: powers of two).  Costs one host read of max |dy| per backward stage -- the drop-in path
    follows the reference's own loop, which synchronises every step anyway (tools/train.py:293); the fused engine scales
    on the device.  Returns None for bf16 / f32 (wide exponent: no scaling)."""
    if rt.dtype_code(dtype) != rt.F16:
        return None
    amax = float(dy.detach().abs().max())
    if not (amax > 0.0) or not math.isfinite(amax):
        return None
    return 2.0 ** (7 - math.frexp(amax)[1])


class EmbedFn(torch.autograd.Function):
    """tokens (B*P, ld) compute dtype -> residual stream (B, P+1, D) fp32:
    rows 1..P = tokens @ W^T + b + pos[1:], row 0 = cls + pos[0]   (models/sit.py:50,70-73)."""

    @staticmethod
    def forward(ctx, tokens, weight, bias, cls_token, pos_embedding, B, P, dtype):
        D, K = weight.shape
        ld = tokens.shape[1]
        wc, _ = ops.stage_weight(weight.detach().contiguous(), dtype, ldc=ld, want_t=False)
        x = torch.empty((B, P + 1, D), dtype=torch.float32, device=tokens.device)
        pos = pos_embedding.detach().reshape(-1, D)[:P + 1].contiguous()
        ops.gemm_nt(tokens, wc, x.view(B * (P + 1), D), dtype, M=B * P, N=D, K=ld, epilogue=ops.EPI_BIAS_RES,
                    bias=bias.detach(), aux=pos, omap=(P, P + 1, 1), auxmap=(P, 0, 1))
        ops.embed_cls_rows(x, cls_token.detach().reshape(-1).contiguous(), pos, B, P + 1, D)
        ctx.save_for_backward(tokens)
        ctx.meta = (B, P, D, K, ld, dtype, tuple(pos_embedding.shape))
        return x

    @staticmethod
    def backward(ctx, dx):
        (tokens,) = ctx.saved_tensors
        B, P, D, K, ld, dtype, pos_shape = ctx.meta
        dx = dx.contiguous()
        dW_pad = torch.zeros((D, ld), dtype=torch.float32, device=dx.device)
        db = torch.zeros((D,), dtype=torch.float32, device=dx.device)
        S = _f16_scale(dx, dtype)
        ops.gemm_wgrad((dx if S is None else dx * S).view(B * (P + 1), D), tokens, dW_pad, dtype, db=db, M=B * P, N=D, K=ld,
                       dymap=(P, P + 1, 1))
        if S is not None:
            dW_pad.mul_(1.0 / S)
            db.mul_(1.0 / S)
        dpos_used = torch.zeros(((P + 1) * D,), dtype=torch.float32, device=dx.device)
        ops.colsum_f32(dx.view(B, (P + 1) * D), dpos_used)
        dpos = torch.zeros(pos_shape, dtype=torch.float32, device=dx.device)
        dpos.view(-1, D)[:P + 1] = dpos_used.view(P + 1, D)
        dcls = dpos_used[:D].clone().view(1, 1, D)
        return None, dW_pad[:, :K].contiguous(), db, dcls, dpos, None, None, None


class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b for fp32 x of shape (..., K): the reach-through entry used by
    models/mpp.py:115 (`transformer.to_patch_embedding[-1](corrupted_batch)`) and `to_original`
    (models/mpp.py:129).  Output fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias, dtype):
        N, K = weight.shape
        lead = x.shape[:-1]
        x2 = x.detach().reshape(-1, K)
        if x2.stride(-1) != 1 or x2.stride(0) % 4 != 0:
            x2 = x2.contiguous()
        ld = ops.pad8(K)
        xc = ops.cast_rows(x2, dtype, ld=ld)
        wc, wt = ops.stage_weight(weight.detach().contiguous(), dtype, ldc=ld, want_t=ctx.needs_input_grad[0])
        y = torch.empty((x2.shape[0], N), dtype=torch.float32, device=x.device)
        ops.gemm_nt(xc, wc, y, dtype, N=N, K=ld, bias=None if bias is None else bias.detach())
        ctx.save_for_backward(xc, wt)
        ctx.meta = (N, K, ld, dtype, lead, bias is not None)
        return y.view(*lead, N)

    @staticmethod
    def backward(ctx, dy):
        xc, wt = ctx.saved_tensors
        N, K, ld, dtype, lead, has_bias = ctx.meta
        dy2 = dy.reshape(-1, N).contiguous()
        S = _f16_scale(dy2, dtype)
        if S is not None:
            dy2 = dy2 * S
        dW_pad = torch.zeros((N, ld), dtype=torch.float32, device=dy.device)
        db = torch.zeros((N,), dtype=torch.float32, device=dy.device) if has_bias else None
        ops.gemm_wgrad(dy2, xc, dW_pad, dtype, db=db, N=N, K=ld)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((dy2.shape[0], K), dtype=torch.float32, device=dy.device)
            ops.gemm_nt(dy2, wt, dx, dtype, N=K, K=N)        # wt = W^T (K, pad8(N)); contraction over N
            dx = dx.view(*lead, K)
        if S is not None:
            dW_pad.mul_(1.0 / S)
            if db is not None:
                db.mul_(1.0 / S)
            if dx is not None:
                dx = dx * (1.0 / S)
        return dx, dW_pad[:, :K].contiguous(), db, None


class EncoderFn(torch.autograd.Function):
    """vit_pytorch.vit.Transformer.forward (models/sit.py:76, models/mpp.py:128) as one call."""

    @staticmethod
    def forward(ctx, x, cfg_tuple, *params):
        B, N, D = x.shape
        dim, depth, heads, mlp_dim, dtype = cfg_tuple
        assert D == dim
        cfg = ops.encoder_cfg(B, N, dim, depth, heads, mlp_dim, dtype)
        ps = [p.detach().contiguous() for p in params]
        per_layer = [ps[11 * i:11 * (i + 1)] for i in range(depth)]
        P = ops.layer_param_array(per_layer)
        acts, scratch = ops.encoder_workspace(cfg, x.device)
        xin = x.detach().contiguous()
        out = torch.empty_like(xin)
        need_grad = any(ctx.needs_input_grad)
        ops.encoder_fwd(cfg, P, xin.view(B * N, D), out.view(B * N, D), acts, scratch, save=need_grad)
        if need_grad:
            ctx.save_for_backward(xin, acts, *ps)
            ctx.cfg_tuple = cfg_tuple
        return out

    @staticmethod
    def backward(ctx, dy):
        xin, acts, *ps = ctx.saved_tensors
        dim, depth, heads, mlp_dim, dtype = ctx.cfg_tuple
        B, N, D = xin.shape
        cfg = ops.encoder_cfg(B, N, dim, depth, heads, mlp_dim, dtype)
        grads = _zeros_like_all(ps)
        P = ops.layer_param_array([ps[11 * i:11 * (i + 1)] for i in range(depth)])
        G = ops.layer_param_array([grads[11 * i:11 * (i + 1)] for i in range(depth)])
        _, scratch = None, torch.empty(rt.lib.sitk_encoder_scratch_bytes(cfg), dtype=torch.uint8, device=dy.device)
        S = _f16_scale(dy, dtype)
        dx = dy.contiguous().clone() if S is None else dy.contiguous() * S
        ops.encoder_bwd(cfg, P, G, xin.view(B * N, D), dx.view(B * N, D), acts, scratch)
        if S is not None:
            dx.mul_(1.0 / S)
            for g in grads:
                g.mul_(1.0 / S)
        return (dx, None, *grads)


class HeadFn(torch.autograd.Function):
    """pool ('cls' / 'mean') -> LayerNorm -> Linear   (models/sit.py:78-82)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w, b, pool_mean):
        B, N, D = x.shape
        xin = x.detach().contiguous()
        args = [t.detach().contiguous() for t in (ln_w, ln_b, w, b)]
        out = ops.head_fwd(xin.view(B * N, D), *args, B, N, D, pool_mean)
        ctx.save_for_backward(xin, *args)
        ctx.pool_mean = pool_mean
        return out

    @staticmethod
    def backward(ctx, dlogits):
        xin, ln_w, ln_b, w, b = ctx.saved_tensors
        B, N, D = xin.shape
        dx = torch.empty_like(xin)
        g = _zeros_like_all((ln_w, ln_b, w, b))
        ops.head_bwd(xin.view(B * N, D), ln_w, ln_b, w, dlogits.contiguous().float(), dx.view(B * N, D), *g, B, N, D,
                     ctx.pool_mean)
        return (dx, *g, None)


# ---- unfused stages: the encoder path with dropout > 0 (models/sit.py:36,57; every reference config uses 0.0) ------------------
class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm(D) (eps 1e-5, affine) over the last dim of an fp32 (..., D) tensor; fp32 out (the 16-bit rounding of the
    normalised rows happens where the fused path rounds them: at the next Linear's operand load)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        D = x.shape[-1]
        x2 = x.detach().reshape(-1, D).contiguous()
        y, mean, rstd = ops.layernorm_fwd(x2, weight.detach().contiguous(), bias.detach().contiguous(), "f32")
        ctx.save_for_backward(x2, mean, rstd, weight.detach())
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, w = ctx.saved_tensors
        D = x2.shape[1]
        dg, db = torch.zeros(D, device=dy.device), torch.zeros(D, device=dy.device)
        dx = ops.layernorm_bwd(dy.reshape(-1, D).contiguous().float(), x2, mean, rstd, w.contiguous(), None, dg, db, "f32")
        return dx.view(ctx.shape), dg, db


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T scale) v for qkv (B, N, 3 H 64) fp32 laid out [q | k | v], each (h d) h-major -> (B, N, H 64) fp32."""

    @staticmethod
    def forward(ctx, qkv, heads, dtype, scale=0.125):
        B, N, _ = qkv.shape
        code = rt.dtype_code(dtype)
        qc = qkv.detach().reshape(B * N, -1).to(rt.torch_dtype(code)).contiguous()
        o, lse = ops.attention_fwd(qc, B, N, heads, float(scale), dtype)
        ctx.save_for_backward(qc, o, lse)
        ctx.meta = (B, N, heads, dtype, float(scale))
        return o.float().view(B, N, heads * 64)

    @staticmethod
    def backward(ctx, d_o):
        qc, o, lse = ctx.saved_tensors
        B, N, heads, dtype, scale = ctx.meta
        S = _f16_scale(d_o, dtype)
        dob = (d_o if S is None else d_o * S).reshape(B * N, -1).to(qc.dtype).contiguous()
        dqkv = ops.attention_bwd(qc, o, dob, lse, B, N, heads, scale, dtype).float()
        if S is not None:
            dqkv = dqkv * (1.0 / S)
        return dqkv.view(B, N, -1), None, None, None


class GeluFn(torch.autograd.Function):
    """exact-erf GELU (nn.GELU()), fp32, libsitk's elementwise kernel."""

    @staticmethod
    def forward(ctx, u):
        u = u.detach().contiguous()
        g = torch.empty_like(u)
        rt.check(rt.lib.sitk_gelu_fwd(u.data_ptr(), g.data_ptr(), u.numel(), rt.stream_ptr()))
        ctx.save_for_backward(u)
        return g

    @staticmethod
    def backward(ctx, dg):
        (u,) = ctx.saved_tensors
        dg = dg.contiguous().float()
        du = torch.empty_like(u)
        rt.check(rt.lib.sitk_gelu_bwd(dg.data_ptr(), u.data_ptr(), du.data_ptr(), u.numel(), rt.stream_ptr()))
        return du


class DropoutResidualFn(torch.autograd.Function):
    """res + Dropout_p(x) (res may be None): nn.Dropout of the reference block followed by its residual add.  The mask comes
    from a device-side Philox stream (`state`: int64[2] = {seed, draws so far}), not from torch's generator: a seeded
    reference run cannot be replayed bit for bit through it (same distribution, other stream)."""

    recorder = None        # tests: a list that receives every mask drawn, in call order

    @staticmethod
    def forward(ctx, x, res, p, state):
        x = x.detach().contiguous()
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        r = None if res is None else res.detach().contiguous()
        rt.check(rt.lib.sitk_dropout_fwd(x.data_ptr(), rt.ptr(r), y.data_ptr(), mask.data_ptr(), x.numel(), float(p),
                                         state.data_ptr(), rt.stream_ptr()))
        ctx.save_for_backward(mask)
        ctx.p, ctx.has_res = float(p), res is not None
        if DropoutResidualFn.recorder is not None:
            DropoutResidualFn.recorder.append(mask.clone())
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous().float()
        dx = torch.empty_like(dy)
        rt.check(rt.lib.sitk_dropout_bwd(dy.data_ptr(), mask.data_ptr(), dx.data_ptr(), dy.numel(), ctx.p, rt.stream_ptr()))
        return dx, (dy if ctx.has_res else None), None, None
