"""CPU oracle for the SiT hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this file, and
only as the checker / the timed CPU baseline.  The product (surface-vision-transformers_amd/) never
imports it and has no CPU fallback.

A plain-PyTorch fp32 restatement of the reference's algorithm, function by function:

  gather_patches / gather_tokens   tools/preprocessing.py:74-84  out[s,c,j,v] = X[s,c,table[v,j]]
  Encoder (+PreNorm/Attention/FeedForward)
                                   vit_pytorch.vit.Transformer as constructed at models/sit.py:57
                                   and called at models/sit.py:76, models/mpp.py:128
  SiT                              models/sit.py:26-82
  MaskedPatchPretraining           models/mpp.py:25-134

PARITY PIN STATUS
  * gather, SiT wrapper (patch embedding, cls/pos, pooling, head) and the whole MPP wrapper
    (mask / swap / replace corruption, to_original, masked MSE): pinned against the reference's own
    Python, imported from /root/reference in the build container by oracle/make_golden.py; vectors
    committed under tests/golden/.
  * Encoder block arithmetic: the reference delegates it to the third-party package `vit-pytorch`
    (requirements.txt:5, unpinned, NOT vendored, NOT installed, no network) and holds no test or
    golden vector for it => **parity unpinned by the reference**.  Restated here from the published
    algorithm of the PreNorm-wrapper generation of vit_pytorch.vit (the only generation whose
    state-dict layout matches utils/utils.py:17-33: layers.i.0.norm / .0.fn.to_qkv (bias-free) /
    .0.fn.to_out.0 / .1.norm / .1.fn.net.0 / .1.fn.net.3, no final norm) and cross-checked against
    an independent implementation, torch.nn.TransformerEncoderLayer(norm_first=True, gelu)
    (tests/test_oracle.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

ICO6_VERTICES = 40962


# --------------------------------------------------------------------------------------------
# a1/a2/a3: patch gather (integer copy, bit exact)
# --------------------------------------------------------------------------------------------
def gather_patches(x_scv, table_pv):
    """tools/preprocessing.py:77-84.  x_scv (S, C, 40962) -> (S, C, P, V);
    table_pv is PATCH-major (P, V): table_pv[j, v] == csv[str(j)][v]."""
    x = np.asarray(x_scv)
    t = np.asarray(table_pv).astype(np.int64)
    S, C, _ = x.shape
    P, V = t.shape
    out = np.zeros((S, C, P, V), dtype=x.dtype)
    for s in range(S):
        for j in range(P):
            out[s, :, j, :] = x[s][:, t[j]]
    return out


def tokens_from_patches(x_bcpv):
    """Rearrange 'b c n v -> b n (v c)' (models/sit.py:49, models/mpp.py:82-83): f = v*C + c."""
    x = np.asarray(x_bcpv)
    B, C, P, V = x.shape
    return np.ascontiguousarray(x.transpose(0, 2, 3, 1)).reshape(B, P, V * C)


def gather_tokens(x_bvc, table_pv):
    """North-star entry: channels-last raw surface (B, 40962, C) -> tokens (B, P, V*C).
    Equals tokens_from_patches(gather_patches(x.transpose(0,2,1), table))."""
    x = np.asarray(x_bvc)
    t = np.asarray(table_pv).astype(np.int64)
    B, _, C = x.shape
    P, V = t.shape
    return x[:, t.reshape(-1), :].reshape(B, P, V * C)


# --------------------------------------------------------------------------------------------
# a6: encoder (vit-pytorch PreNorm generation)
# --------------------------------------------------------------------------------------------
class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)  # eps 1e-5, affine, biased variance
        self.fn = fn

    def forward(self, x):
        return self.fn(self.norm(x))


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.0):
        super().__init__()
        # indices 0 and 3 carry the weights (utils/utils.py:29-33)
        self.net = nn.Sequential(
            nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
            nn.Linear(hidden_dim, dim), nn.Dropout(dropout))

    def forward(self, x):
        return self.net(x)


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        self.to_qkv = nn.Linear(dim, 3 * inner, bias=False)  # weight only (utils/utils.py:24)
        project_out = not (heads == 1 and dim_head == dim)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout)) if project_out else nn.Identity()

    def forward(self, x):
        B, N, _ = x.shape
        H, dh = self.heads, self.dim_head
        q, k, v = self.to_qkv(x).chunk(3, dim=-1)                      # order q | k | v
        q, k, v = (t.reshape(B, N, H, dh).permute(0, 2, 1, 3) for t in (q, k, v))  # (h d) h-major
        dots = torch.matmul(q, k.transpose(-1, -2)) * self.scale       # scale AFTER the product
        attn = dots.softmax(dim=-1)
        out = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape(B, N, H * dh)
        return self.to_out(out)


class Encoder(nn.Module):
    """Positional ctor order (dim, depth, heads, dim_head, mlp_dim, dropout) as used at
    models/sit.py:57.  No final norm."""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.0):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([PreNorm(dim, Attention(dim, heads, dim_head, dropout)),
                           PreNorm(dim, FeedForward(dim, mlp_dim, dropout))])
            for _ in range(depth)])

    def forward(self, x):
        for attn, ff in self.layers:
            x = attn(x) + x
            x = ff(x) + x
        return x


# --------------------------------------------------------------------------------------------
# a3-a7: SiT
# --------------------------------------------------------------------------------------------
class _ToTokens(nn.Module):
    def forward(self, x):  # (B, C, P, V) -> (B, P, V*C)
        B, C, P, V = x.shape
        return x.permute(0, 2, 3, 1).reshape(B, P, V * C)


class SiT(nn.Module):
    """models/sit.py:26-82 with the encoder above in place of vit_pytorch.vit.Transformer."""

    def __init__(self, *, dim, depth, heads, mlp_dim, pool="cls", num_patches=20, num_classes=1,
                 num_channels=4, num_vertices=2145, dim_head=64, dropout=0.0, emb_dropout=0.0):
        super().__init__()
        assert pool in {"cls", "mean"}, "pool type must be either cls (cls token) or mean (mean pooling)"
        patch_dim = num_channels * num_vertices
        self.to_patch_embedding = nn.Sequential(_ToTokens(), nn.Linear(patch_dim, dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Encoder(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))

    def embed_tokens(self, tokens):
        """tokens (B, P, K) -> residual stream (B, P+1, D): Linear, cls cat, +pos, dropout."""
        x = self.to_patch_embedding[1](tokens)
        B, n, _ = x.shape
        x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)
        x = x + self.pos_embedding[:, :n + 1]
        return self.dropout(x)

    def forward(self, img):
        x = self.embed_tokens(self.to_patch_embedding[0](img))
        x = self.transformer(x)
        x = x.mean(dim=1) if self.pool == "mean" else x[:, 0]
        return self.mlp_head(self.to_latent(x))


# --------------------------------------------------------------------------------------------
# a9/a10: masked patch pre-training
# --------------------------------------------------------------------------------------------
def draw_mpp_randoms(B, P, mask_prob, replace_prob, swap_prob, device="cpu"):
    """The four random tensors of models/mpp.py in the reference's draw order
    (rand -> uniform_ -> randint -> uniform_; mpp.py:34, :43 via :94, :99, :43 via :109).
    `rand`/`randint` use `device`'s generator, the two uniform_ draws always the CPU generator."""
    n_mask = math.ceil(mask_prob * P)
    r = torch.rand((B, P), device=device)
    idx = r.topk(n_mask, dim=-1).indices
    corrupted = torch.zeros((B, P), device=device).scatter_(1, idx, 1).bool()
    out = {"corrupted_sequence": corrupted}
    if swap_prob > 0:
        p = swap_prob / (1 - replace_prob)
        out["swap_draw"] = (torch.zeros((B, P)).float().uniform_(0, 1) < p).to(device)
        out["random_patches"] = torch.randint(0, P, (B, P), device=device)
    out["replace_draw"] = (torch.zeros((B, P)).float().uniform_(0, 1) < replace_prob).to(device)
    return out


class MaskedPatchPretraining(nn.Module):
    """models/mpp.py:46-134.  `randoms` (see draw_mpp_randoms) may be injected to replay a
    captured reference run; when None they are drawn in the reference's order."""

    def __init__(self, transformer, dim_in, dim_out, device="cpu", mask_prob=0.15, replace_prob=0.5,
                 swap_prob=0.3, channels=4, num_vertices=561):
        super().__init__()
        self.transformer = transformer
        self.dim_in, self.dim_out = dim_in, dim_out
        self.to_original = nn.Linear(dim_in, dim_out).to(device)
        self.mask_prob, self.replace_prob, self.swap_prob = mask_prob, replace_prob, swap_prob
        self.mask_token = nn.Parameter(torch.randn(1, 1, channels * num_vertices))

    def corrupt(self, tokens, randoms):
        B, P, _ = tokens.shape
        masked = randoms["corrupted_sequence"]
        corrupted = tokens.clone().detach()
        if self.swap_prob > 0:
            swap = masked & randoms["swap_draw"]
            src = corrupted[torch.arange(B).unsqueeze(-1), randoms["random_patches"]]  # from the clean clone
            corrupted = torch.where(swap.unsqueeze(-1), src, corrupted)
        replace = masked & randoms["replace_draw"]                      # applied second: wins over swap
        return torch.where(replace.unsqueeze(-1), self.mask_token.expand(B, P, -1), corrupted)

    def forward(self, batch, randoms=None):
        sit = self.transformer
        tokens = sit.to_patch_embedding[0](batch)
        B, P, K = tokens.shape
        if randoms is None:
            randoms = draw_mpp_randoms(B, P, self.mask_prob, self.replace_prob, self.swap_prob, tokens.device)
        masked = randoms["corrupted_sequence"]
        x = sit.embed_tokens(self.corrupt(tokens, randoms))
        x = sit.transformer(x)
        batch_out = self.to_original(x[:, 1:, :])
        mpp_loss = F.mse_loss(batch_out[masked], tokens[masked])       # mean over B*ceil(p*P)*K elems
        return mpp_loss, batch_out


MODEL_SIZES = {  # config/SiT/training/hparams.yml:33-45
    "tiny": dict(dim=192, depth=12, heads=3, mlp_dim=768, dim_head=64),
    "small": dict(dim=384, depth=12, heads=6, mlp_dim=1536, dim_head=64),
    "base": dict(dim=768, depth=12, heads=12, mlp_dim=3072, dim_head=64),
}
