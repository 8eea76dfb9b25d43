"""Micro-benchmark of single hot-path kernels at the headline shapes (for rocprofv3 --pmc runs).

    python tools/kbench.py <kernel> [--reps 20] [--batch 64]
kernels: wgrad_w1 wgrad_w2 wgrad_qkv wgrad_layer gemm_qkv gemm_fc1 gemm_fc2 gemm_dfc2 gemm_dfc1 attn_fwd attn_bwd attn_bwd_do attn_bwd_proj ln_fwd ln_bwd mlp_fwd mlp_fwd_nosave mlp_bwd lnqkv_fwd lnqkv_bwd lnqkv_mlp_bwd
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import ops  # noqa: E402


WS = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dim", type=int, default=192)
    ap.add_argument("--heads", type=int, default=3)
    ap.add_argument("--tokens", type=int, default=321)
    a = ap.parse_args()
    dev, dt, td = "cuda:0", "bf16", torch.bfloat16
    B, N, D, H = a.batch, a.tokens, a.dim, a.heads
    M, I, R = 4 * D, H * 64, B * N
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, dtype=td: (torch.randn(*s, device=dev, generator=g) * 0.5).to(dtype)  # noqa: E731
    f32 = torch.float32
    h, qkv, o, u = rn(R, D), rn(R, 3 * I), rn(R, I), rn(R, M)
    x32 = rn(R, D, dtype=f32)
    dxT = rn(R, D)
    w = {k: rn(*s) for k, s in dict(qkv=(3 * I, D), qkv_t=(D, 3 * I), w1=(M, D), w1_t=(D, M), w2=(D, M), w2_t=(M, D)).items()}
    bD, bM = rn(D, dtype=f32), rn(M, dtype=f32)
    dW = {k: torch.zeros(s, dtype=f32, device=dev) for k, s in dict(qkv=(3 * I, D), o=(D, I), w1=(M, D), w2=(D, M)).items()}
    out_qkv, out_u, out_g, out_h, out_x = torch.empty_like(qkv), torch.empty_like(u), torch.empty_like(u), torch.empty_like(h), torch.empty_like(x32)
    o_att, lse = ops.attention_fwd(qkv, B, N, H, 0.125, dt)
    wo_t, out_o = rn(I, D), torch.empty_like(o)
    mean, rstd = torch.zeros(R, device=dev), torch.ones(R, device=dev)
    fns = {
        "wgrad_w1": lambda: ops.gemm_wgrad(u, h, dW["w1"], dt, db=bM),
        "wgrad_w2": lambda: ops.gemm_wgrad(dxT, u, dW["w2"], dt, db=bD),
        "wgrad_qkv": lambda: ops.gemm_wgrad(qkv, h, dW["qkv"], dt),
        "wgrad_layer": lambda: ops.gemm_wgrad_group([
            dict(dY=dxT, X=u, dW=dW["w2"], db=bD), dict(dY=u, X=h, dW=dW["w1"], db=bM),
            dict(dY=dxT, X=o, dW=dW["o"], db=bD), dict(dY=qkv, X=h, dW=dW["qkv"])], dt, workspace=WS),
        "wgrad_12layers": lambda: ops.gemm_wgrad_group([
            dict(dY=dxT, X=u, dW=dW["w2"], db=bD), dict(dY=u, X=h, dW=dW["w1"], db=bM),
            dict(dY=dxT, X=o, dW=dW["o"], db=bD), dict(dY=qkv, X=h, dW=dW["qkv"])] * 12, dt, workspace=WS),
        "gemm_qkv": lambda: ops.gemm_nt(h, w["qkv"], out_qkv, dt),
        "gemm_fc1": lambda: ops.gemm_nt(h, w["w1"], out_u, dt, epilogue=ops.EPI_BIAS_GELU, bias=bM, out2=out_g),
        "gemm_fc2": lambda: ops.gemm_nt(u, w["w2"], out_x, dt, epilogue=ops.EPI_BIAS_RES, bias=bD, aux=x32),
        "gemm_dfc2": lambda: ops.gemm_nt(dxT, w["w2_t"], out_u, dt, epilogue=ops.EPI_DGELU, aux=u),
        "gemm_dfc1": lambda: ops.gemm_nt(u, w["w1_t"], out_h, dt),
        "attn_fwd": lambda: ops.attention_fwd(qkv, B, N, H, 0.125, dt),
        "attn_bwd": lambda: ops.attention_bwd(qkv, o_att, o, lse, B, N, H, 0.125, dt),
        "attn_bwd_do": lambda: (ops.gemm_nt(dxT, wo_t, out_o, dt), ops.attention_bwd(qkv, o_att, out_o, lse, B, N, H, 0.125, dt)),
        "attn_bwd_proj": lambda: ops.attention_bwd_proj(qkv, o_att, dxT, wo_t, lse, B, N, H, 0.125, dt),
        "mlp_fwd": lambda: ops.mlp_fwd(x32, bD, bD, w["w1"], bM, w["w2"], bD, dt),
        "mlp_fwd_nosave": lambda: ops.mlp_fwd(x32, bD, bD, w["w1"], bM, w["w2"], bD, dt, save=False),
        "mlp_bwd": lambda: ops.mlp_bwd(x32, dxT, x32, mean, rstd, bD, w["w2_t"], w["w1_t"], u, dt),
        "lnqkv_fwd": lambda: ops.ln_gemm_fwd(x32, bD, bD, w["qkv"], dt),
        "lnqkv_bwd": lambda: ops.ln_gemm_bwd(qkv, w["qkv_t"], x32, mean, rstd, bD, x32, dt),
        "lnqkv_mlp_bwd": lambda: ops.ln_gemm_mlp_bwd(qkv, w["qkv_t"], x32, mean, rstd, bD, x32, x32, mean, rstd, bD, w["w2_t"], w["w1_t"], u, dt),
        "proj_mlp_fwd": lambda: ops.attn_out_mlp_fwd(o, w["qkv"][:D].contiguous(), bD, x32, bD, bD, w["w1"], bM, w["w2"], bD, dt, want_g=True),
        "proj_mlp_next_fwd": lambda: ops.attn_out_mlp_next_fwd(o, w["qkv"][:D].contiguous(), bD, x32, bD, bD, w["w1"], bM, w["w2"], bD, bD, bD, w["qkv"], dt, want_g=True),
        "mlp_fwd_g": lambda: ops.mlp_fwd(x32, bD, bD, w["w1"], bM, w["w2"], bD, dt, want_g=True),
        "proj": lambda: ops.gemm_nt(o, w["qkv"][:D].contiguous(), out_x, dt, epilogue=ops.EPI_BIAS_RES, bias=bD, aux=x32),
        "tiny": lambda: ops.layernorm_fwd(x32[:128], bD, bD, dt),
        "ln_fwd": lambda: ops.layernorm_fwd(x32, bD, bD, dt),
        "ln_bwd": lambda: ops.layernorm_bwd(h, x32, mean, rstd, bD, x32, bD.clone(), bD.clone(), dt, dx=out_x),
    }
    global WS
    WS = torch.empty(128 << 20, dtype=torch.uint8, device=dev)
    fn = fns[a.kernel]
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    # GPU-paced timing: the launches are replayed from one hipGraph (host launch cost of the python wrappers
    # would otherwise dominate kernels shorter than ~20 us)
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for _ in range(a.reps):
                fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{a.kernel}: {e0.elapsed_time(e1) / (5 * a.reps) * 1e3:.1f} us/call (graph replay of {a.reps} launches)")


if __name__ == "__main__":
    main()
