"""Attention kernels by themselves at a given shape: forward, backward query side, backward key side (device time per
launch from a hipGraph of launches over rotating buffer sets; algorithmic TFLOP/s: forward 4 B H N^2 64, each backward
kernel the same -- SURVEY 8(d): backward = 2 x forward).

    python tools/attn_bench.py [--batch 32 --tokens 1281 --heads 6 --sets 4 --reps 8]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import ops  # noqa: E402
from sitk import runtime as rt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=1281)
    ap.add_argument("--heads", type=int, default=6)
    ap.add_argument("--sets", type=int, default=4)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev, dt = "cuda:0", "bf16"
    B, N, H = a.batch, a.tokens, a.heads
    I, R = H * 64, B * N
    g = torch.Generator(device=dev).manual_seed(0)
    S = []
    for _ in range(a.sets):
        qkv = (torch.randn(R, 3 * I, device=dev, generator=g)).to(torch.bfloat16)
        d_o = (torch.randn(R, I, device=dev, generator=g)).to(torch.bfloat16)
        o, lse = ops.attention_fwd(qkv, B, N, H, 0.125, dt)
        S.append(dict(qkv=qkv, d_o=d_o, o=o, lse=lse, delta=torch.zeros_like(lse), dqkv=torch.empty_like(qkv),
                      o2=torch.empty_like(o), lse2=torch.empty_like(lse)))

    def fwd(s):
        rt.check(rt.lib.sitk_attention_fwd(s["qkv"].data_ptr(), s["o2"].data_ptr(), s["lse2"].data_ptr(), B, N, H, 0.125, rt.BF16,
                                           rt.stream_ptr()))

    def bwd(s, phases):
        rt.check(rt.lib.sitk_attention_bwd_phases(s["qkv"].data_ptr(), s["o"].data_ptr(), s["d_o"].data_ptr(), None, None, None,
                                                  s["lse"].data_ptr(), s["delta"].data_ptr(), s["dqkv"].data_ptr(), B, N, H, I,
                                                  0.125, rt.BF16, phases, rt.stream_ptr()))
    flops = 4.0 * B * H * N * N * 64
    for name, fn in (("fwd", fwd), ("bwd_dq", lambda s: bwd(s, 1)), ("bwd_dkv", lambda s: bwd(s, 2))):
        if a.only and a.only != name:
            continue
        for s in S:
            fn(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(a.reps):
                fn(S[i % a.sets])
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (3 * a.reps) * 1e3
        print(f"{name:8s} B={B} N={N} H={H}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  ({flops / us / 1e6 / 2500 * 100:.1f} % of the bf16 MFMA peak)",
              flush=True)


if __name__ == "__main__":
    main()
