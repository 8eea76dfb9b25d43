"""Masked patch pre-training on the MI355X-native kernels.

Host-side mirror of the reference's models/mpp.py:46-134: same constructor, same attributes
(`transformer`, `to_original`, `mask_token`, the three probabilities), same state-dict keys,
`forward(batch) -> (mpp_loss, batch_out)` with batch (B, C, P, V) [or a raw (B, 40962, C) surface].

Corruption semantics (models/mpp.py:85-112): exactly ceil(mask_prob * P) tokens per sample are
selected; a selected token is swapped with a random token of the SAME sample's clean sequence with
probability swap_prob / (1 - replace_prob), then replaced by `mask_token` with probability
replace_prob (replacement wins).  The random tensors are drawn with torch in the reference's order
and from the same generators (rand -> uniform_(CPU) -> randint -> uniform_(CPU)) so a seeded run
replays; they can also be injected (`randoms=`) to replay masks captured elsewhere.
The loss is the reference's mean squared error over the masked rows (models/mpp.py:132), evaluated
densely (no boolean-index gathers, hence no device synchronisation).
"""
import math

import torch
from torch import nn

from .. import functional as Fn
from .. import ops
from ..runtime import SitkError


def draw_randoms(B, P, mask_prob, replace_prob, swap_prob, device):
    n_mask = math.ceil(mask_prob * P)
    scores = torch.rand((B, P), device=device)
    picked = scores.topk(n_mask, dim=-1).indices
    masked = torch.zeros((B, P), device=device).scatter_(1, picked, 1).bool()
    out = {"corrupted_sequence": masked}
    if swap_prob > 0:
        p_swap = swap_prob / (1 - replace_prob)
        out["swap_draw"] = (torch.zeros((B, P)).float().uniform_(0, 1) < p_swap).to(device)
        out["random_patches"] = torch.randint(0, P, (B, P), device=device)
    out["replace_draw"] = (torch.zeros((B, P)).float().uniform_(0, 1) < replace_prob).to(device)
    return out


class _MppEmbedFn(torch.autograd.Function):
    """EmbedFn + gradient of `mask_token`: d mask_token = (sum over replaced rows of dY) @ W_embed
    (linearity), i.e. a masked column sum and a 1-row GEMM instead of a (B*P, K) input gradient."""

    @staticmethod
    def forward(ctx, tokens, weight, bias, cls_token, pos_embedding, mask_token, replaced_full, B, P, dtype):
        x = Fn.EmbedFn.forward(ctx, tokens, weight, bias, cls_token, pos_embedding, B, P, dtype)
        ctx.mpp = (weight.detach(), replaced_full, tuple(mask_token.shape))
        return x

    @staticmethod
    def backward(ctx, dx):
        g = Fn.EmbedFn.backward(ctx, dx)
        weight, replaced_full, mt_shape = ctx.mpp
        B, P, D, K, ld, dtype, _ = ctx.meta
        dx2 = dx.contiguous().view(B * (P + 1), D)
        r = torch.zeros((1, D), dtype=torch.float32, device=dx.device)
        ops.masked_colsum(dx2, replaced_full, None, r, "f32")
        _, wt = ops.stage_weight(weight.contiguous(), dtype, want_c=False)      # (K, pad8(D)) = W^T
        dmt = torch.empty((1, K), dtype=torch.float32, device=dx.device)
        S = Fn._f16_scale(r, dtype)                                             # f16 mode: scaled operand, see functional.py
        ops.gemm_nt(r if S is None else r * S, wt, dmt, dtype, M=1, N=K, K=D)
        if S is not None:
            dmt.mul_(1.0 / S)
        return g[0], g[1], g[2], g[3], g[4], dmt.view(mt_shape), None, None, None, None


class _MppLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, batch_out, tokens, masked_u8, n_masked_total):
        rows = batch_out.shape[0] * batch_out.shape[1]
        K = batch_out.shape[2]
        loss = torch.zeros((1,), dtype=torch.float32, device=batch_out.device)
        dout = torch.empty_like(batch_out)
        ops.mpp_loss_fwd_bwd(batch_out.detach().contiguous().view(rows, K), tokens, masked_u8, loss, dout.view(rows, K),
                             n_masked_total)
        ctx.save_for_backward(dout)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dout,) = ctx.saved_tensors
        return dout * g, None, None, None


class masked_patch_pretraining(nn.Module):  # noqa: N801  (name fixed by tools/pretrain.py:246)

    def __init__(self, transformer, dim_in, dim_out, device, mask_prob=0.15, replace_prob=0.5, swap_prob=0.3,
                 channels=4, num_vertices=561):
        super().__init__()
        self.transformer = transformer
        self.dim_in, self.dim_out = dim_in, dim_out
        self.to_original = nn.Linear(dim_in, dim_out)
        self.to_original.to(device)
        self.mask_prob, self.replace_prob, self.swap_prob = mask_prob, replace_prob, swap_prob
        self.mask_token = nn.Parameter(torch.randn(1, 1, channels * num_vertices))

    def forward(self, batch, randoms=None, **kwargs):
        sit = self.transformer
        if not batch.is_cuda:
            raise SitkError("sitk masked_patch_pretraining: input must be on the GPU (no CPU path)")
        dtype = sit.compute_dtype
        P, K = sit.num_patches, sit.patch_dim
        B = batch.shape[0]
        dev = batch.device
        # clean tokens, fp32: the regression target (models/mpp.py:82-83,132)
        if batch.dim() == 4:
            tokens = ops.patchify(batch.float(), "f32", ld=ops.pad8(K))
        else:
            tokens = ops.gather_tokens(batch.float().contiguous(), sit.patch_table(dev), "f32", ld=ops.pad8(K))
        if tokens.shape[1] != K:
            tokens = tokens[:, :K].contiguous()
        if randoms is None:
            randoms = draw_randoms(B, P, self.mask_prob, self.replace_prob, self.swap_prob, dev)
            n_masked = B * math.ceil(self.mask_prob * P)          # exact by construction, no device sync
        else:
            n_masked = int(randoms["corrupted_sequence"].sum())  # replayed masks may have any count
        masked = randoms["corrupted_sequence"].to(dev)
        u8 = lambda t: t.to(dev).reshape(-1).to(torch.uint8).contiguous()  # noqa: E731
        swap = u8(randoms["swap_draw"]) if self.swap_prob > 0 else None
        rpatch = randoms["random_patches"].to(dev).reshape(-1).to(torch.int32).contiguous() if self.swap_prob > 0 else None
        replace_draw = randoms["replace_draw"].to(dev)
        mask_token = self.mask_token.to(dev)
        corrupted = ops.mpp_corrupt(tokens, u8(masked), swap, rpatch, u8(replace_draw),
                                    mask_token.detach().reshape(-1).contiguous(), B, P, K, dtype)
        replaced_full = torch.zeros((B, P + 1), dtype=torch.uint8, device=dev)
        replaced_full[:, 1:] = (masked & replace_draw).to(torch.uint8)

        lin = sit.to_patch_embedding[-1]
        x = _MppEmbedFn.apply(corrupted, lin.weight, lin.bias, sit.cls_token, sit.pos_embedding, mask_token,
                              replaced_full.reshape(-1), B, P, dtype)
        x = sit.dropout(x)
        x = sit.transformer(x, **kwargs)
        batch_out = Fn.LinearFn.apply(x[:, 1:, :], self.to_original.weight, self.to_original.bias, dtype)
        mpp_loss = _MppLossFn.apply(batch_out, tokens, u8(masked), int(n_masked))
        return mpp_loss, batch_out
