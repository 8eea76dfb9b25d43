"""Round 6 (review item 4, the structural attempt at the tiny chain): do two HALF-batch kernel chains on two streams, out of
phase, finish sooner than the one whole-batch chain?

Every chain kernel of the tiny step is ONE wave of workgroups bound by one workgroup's critical path (a half batch takes ~80 % of
the whole batch's time), attention leaves 64 CUs idle and the fused MLP kernels 42, and attention (vector / LDS bound, 0.2 of HBM)
and the fused MLP kernels (HBM bound) want different resources.  Two chains over the two halves of the batch -- the SAME kernels on
row ranges [0, R/2) and [R/2, R) of the same buffers (10 272 = 107 x 96 rows, 32 samples: both partitions divide) -- would run
attention of one half beside the MLP kernel of the other.  Rounds 2 / 3 measured "no gain" (tools/dual_stream.py: the SAME kernel
in phase, which contends with itself) and "slower" (tools/dual_stream2.py: eager launches of 2 x 18 kernels through Python -- the
host paces that run -- and the second chain on the LOWEST-priority side stream).  This tool replays each chain from a hipGraph
(GPU-paced), on two default-priority streams whose placement passed the engine's probe (no toxic hardware queue), the second
chain rotated by one kernel so that the two are out of phase.

    python tools/dual_chain.py [--layers 12]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import engine, ops  # noqa: E402

dev, dt, td = "cuda:0", "bf16", torch.bfloat16
N, D, H = 321, 192, 3
M, I = 4 * D, H * 64


def make(B):
    R = B * N
    g = torch.Generator(device=dev).manual_seed(B)
    rn = lambda *s, dtype=td: (torch.randn(*s, device=dev, generator=g) * 0.5).to(dtype)  # noqa: E731
    t = dict(R=R, B=B, x32=rn(R, D, dtype=torch.float32), w1=rn(M, D), w2=rn(D, M), w1t=rn(D, M), w2t=rn(M, D),
             bM=rn(M, dtype=torch.float32), bD=rn(D, dtype=torch.float32), u=rn(R, M), dxc=rn(R, D), qkv=rn(R, 3 * I),
             o=rn(R, I), wqkv=rn(3 * I, D), wqkv_t=rn(D, 3 * I), wo_t=rn(I, D), wo=rn(D, I), mean=torch.zeros(R, device=dev),
             rstd=torch.ones(R, device=dev))
    t["o_att"], t["lse"] = ops.attention_fwd(t["qkv"], B, N, H, 0.125, dt)
    return t


def fwd_kernels(t):
    return [lambda: ops.attention_fwd(t["qkv"], t["B"], N, H, 0.125, dt),
            lambda: ops.attn_out_mlp_next_fwd(t["o"], t["wo"], t["bD"], t["x32"], t["bD"], t["bD"], t["w1"], t["bM"], t["w2"], t["bD"],
                                              t["bD"], t["bD"], t["wqkv"], dt, want_g=True)]


def bwd_kernels(t):
    return [lambda: ops.attention_bwd_proj(t["qkv"], t["o_att"], t["dxc"], t["wo_t"], t["lse"], t["B"], N, H, 0.125, dt),
            lambda: ops.ln_gemm_mlp_bwd(t["qkv"], t["wqkv_t"], t["x32"], t["mean"], t["rstd"], t["bD"], t["x32"], t["x32"], t["mean"],
                                        t["rstd"], t["bD"], t["w2t"], t["w1t"], t["u"], dt)]


def graph_of(kernels, layers, stream, rotate=0):
    seq = [kernels[(i + rotate) % len(kernels)] for i in range(layers * len(kernels))]
    for k in kernels:
        k()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(g, stream=stream):
            for k in seq:
                k()
    return g


def timed(fn, reps=12):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=12)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    s1, p1 = engine.pick_bucket_stream(torch.device(dev))
    s2, p2 = engine.pick_bucket_stream(torch.device(dev))
    print("streams:", [(r["stream"], round(r["blocked_us"] / r["free_us"], 2), r["ok"]) for r in p1 + p2 if r["chosen"]])
    full, ha, hb = make(64), make(32), make(32)
    cur = torch.cuda.current_stream()
    for name, kernels in (("forward [attention, block tail]", fwd_kernels), ("backward [attention, d to_qkv + MLP]", bwd_kernels)):
        gf = graph_of(kernels(full), a.layers, s1)
        ga = graph_of(kernels(ha), a.layers, s1)
        for rotate in (0, 1):
            gb = graph_of(kernels(hb), a.layers, s2, rotate=rotate)

            def whole():
                s1.wait_stream(cur)
                with torch.cuda.stream(s1):
                    gf.replay()
                cur.wait_stream(s1)

            def dual():
                s1.wait_stream(cur)
                s2.wait_stream(cur)
                with torch.cuda.stream(s1):
                    ga.replay()
                with torch.cuda.stream(s2):
                    gb.replay()
                cur.wait_stream(s1)
                cur.wait_stream(s2)

            def serial():
                s1.wait_stream(cur)
                with torch.cuda.stream(s1):
                    ga.replay()
                    ga.replay()
                cur.wait_stream(s1)

            for rep in range(2):
                print(f"{name}, {a.layers} layers: whole batch {timed(whole):7.0f} us   two half chains, second one rotated by {rotate}: "
                      f"{timed(dual):7.0f} us   (one half chain twice in a row {timed(serial):7.0f} us)", flush=True)


if __name__ == "__main__":
    main()
