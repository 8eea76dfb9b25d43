"""Experiment (round 3): two HALF batches as two free-running kernel chains -- the main stream and a library-owned side stream
(no hardware-queue aliasing with torch's pool streams) -- against one whole-batch chain.  The chains are the backward kernels of
one layer (MLP backward, attention backward pair, to_qkv backward), 6 layers deep; the second chain starts `offset` kernels late
so that the two are out of phase.  Every kernel is latency-bound by one workgroup's critical path (a half batch takes ~80 % of the
whole batch's time alone): do two de-phased half chains finish sooner than the whole chain?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import ops, runtime as rt  # noqa: E402

dev, dt, td = "cuda:0", "bf16", torch.bfloat16
N, D, H = 321, 192, 3
M, I = 4 * D, H * 64


def make(B):
    R = B * N
    g = torch.Generator(device=dev).manual_seed(B)
    rn = lambda *s, dtype=td: (torch.randn(*s, device=dev, generator=g) * 0.5).to(dtype)  # noqa: E731
    t = dict(R=R, B=B, x32=rn(R, D, dtype=torch.float32), w1=rn(M, D), w2=rn(D, M), w1t=rn(D, M), w2t=rn(M, D),
             bM=rn(M, dtype=torch.float32), bD=rn(D, dtype=torch.float32), u=rn(R, M), dxc=rn(R, D), qkv=rn(R, 3 * I),
             o=rn(R, I), wqkv=rn(3 * I, D), wqkv_t=rn(D, 3 * I), wo_t=rn(I, D), mean=torch.zeros(R, device=dev),
             rstd=torch.ones(R, device=dev))
    t["o_att"], t["lse"] = ops.attention_fwd(t["qkv"], B, N, H, 0.125, dt)
    return t


def layer_bwd(t):
    ops.mlp_bwd(t["x32"], t["dxc"], t["x32"], t["mean"], t["rstd"], t["bD"], t["w2t"], t["w1t"], t["u"], dt)
    ops.attention_bwd_proj(t["qkv"], t["o_att"], t["dxc"], t["wo_t"], t["lse"], t["B"], N, H, 0.125, dt)
    ops.ln_gemm_bwd(t["qkv"], t["wqkv_t"], t["x32"], t["mean"], t["rstd"], t["bD"], t["x32"], dt)


def layer_fwd(t):
    ops.attention_fwd(t["qkv"], t["B"], N, H, 0.125, dt)
    ops.attn_out_mlp_next_fwd(t["o"], t["wqkv"][:D].contiguous(), t["bD"], t["x32"], t["bD"], t["bD"], t["w1"], t["bM"], t["w2"], t["bD"],
                              t["bD"], t["bD"], t["wqkv"], dt, want_g=True) if t["R"] <= 24576 else None


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


full, ha, hb = make(64), make(32), make(32)
ov = rt.lib.sitk_overlap_create(1, 42, 1)
side = torch.cuda.ExternalStream(rt.lib.sitk_overlap_stream(ov))
main = torch.cuda.current_stream()
L = 6
for name, layer in (("backward", layer_bwd), ("forward", layer_fwd)):
    for _ in range(2):
        layer(full); layer(ha); layer(hb)

    def whole():
        for _ in range(L):
            layer(full)

    def halves_serial():
        for _ in range(L):
            layer(ha)
            layer(hb)

    def dual(offset):
        def run():
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(offset):          # a short delay: the second chain starts out of phase
                    ops.layernorm_fwd(hb["x32"], hb["bD"], hb["bD"], dt)
                for _ in range(L):
                    layer(hb)
            for _ in range(L):
                layer(ha)
            main.wait_stream(side)
        return run

    for rep in range(3):
        print(f"{name}, {L} layers, us: whole batch {timed(whole):.0f}   half batches one after the other {timed(halves_serial):.0f}   "
              f"two streams, in phase {timed(dual(0)):.0f}   second stream 1 / 3 short kernels late {timed(dual(1)):.0f} / {timed(dual(3)):.0f}")
rt.lib.sitk_overlap_destroy(ov)
