"""Experiment: the same kernel on two half batches in two concurrent streams vs once on the whole batch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import ops  # noqa: E402

dev, dt, td = "cuda:0", "bf16", torch.bfloat16
N, D, H = 321, 192, 3
M, I = 4 * D, H * 64


def make(B):
    R = B * N
    g = torch.Generator(device=dev).manual_seed(B)
    rn = lambda *s, dtype=td: (torch.randn(*s, device=dev, generator=g) * 0.5).to(dtype)  # noqa: E731
    t = dict(R=R, B=B, x32=rn(R, D, dtype=torch.float32), w1=rn(M, D), w2=rn(D, M), w1t=rn(D, M), w2t=rn(M, D),
             bM=rn(M, dtype=torch.float32), bD=rn(D, dtype=torch.float32), u=rn(R, M), dxc=rn(R, D), qkv=rn(R, 3 * I),
             o=rn(R, I), wqkv=rn(3 * I, D), mean=torch.zeros(R, device=dev), rstd=torch.ones(R, device=dev))
    t["o_att"], t["lse"] = ops.attention_fwd(t["qkv"], B, N, H, 0.125, dt)
    return t


def kern(name, t):
    if name == "mlp_fwd":
        return lambda: ops.mlp_fwd(t["x32"], t["bD"], t["bD"], t["w1"], t["bM"], t["w2"], t["bD"], dt, want_g=True)
    if name == "mlp_bwd":
        return lambda: ops.mlp_bwd(t["x32"], t["dxc"], t["x32"], t["mean"], t["rstd"], t["bD"], t["w2t"], t["w1t"], t["u"], dt)
    if name == "lnqkv_fwd":
        return lambda: ops.ln_gemm_fwd(t["x32"], t["bD"], t["bD"], t["wqkv"], dt)
    if name == "attn_fwd":
        return lambda: ops.attention_fwd(t["qkv"], t["B"], N, H, 0.125, dt)
    if name == "attn_bwd":
        return lambda: ops.attention_bwd(t["qkv"], t["o_att"], t["o"], t["lse"], t["B"], N, H, 0.125, dt)
    raise KeyError(name)


def timed(fn, reps=30):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(reps)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


full, ha, hb = make(64), make(32), make(32)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name in ["mlp_fwd", "mlp_bwd", "lnqkv_fwd", "attn_fwd", "attn_bwd"]:
    f, a, b = kern(name, full), kern(name, ha), kern(name, hb)
    # graphs of 10 launches per stream (no host pacing); the two half-batch graphs replay on two streams
    def graph_of(fn, stream):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            with torch.cuda.graph(g, stream=stream):
                for _ in range(10):
                    fn()
        return g
    gf, ga, gb = graph_of(f, s1), graph_of(a, s1), graph_of(b, s2)

    def run_full(reps):
        for _ in range(reps):
            gf.replay()

    def run_dual(reps):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        for _ in range(reps):
            with torch.cuda.stream(s1):
                ga.replay()
            with torch.cuda.stream(s2):
                gb.replay()
        cur.wait_stream(s1); cur.wait_stream(s2)

    tf = timed(run_full) / 10
    td2 = timed(run_dual) / 10
    print(f"{name}: whole batch {tf:.1f} us/launch; two half batches on two streams {td2:.1f} us per pair")
