"""The encoder's forward / input-gradient GEMMs by themselves at one model size (device time per launch from a hipGraph over
rotating buffer sets; algorithmic TFLOP/s and GB/s).

    python tools/gemm_bench.py [--model small|base|tiny --tokens 40992 --sets 4]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import ops  # noqa: E402

DIMS = {"tiny": (192, 3), "small": (384, 6), "base": (768, 12)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="small")
    ap.add_argument("--tokens", type=int, default=32 * 1281)
    ap.add_argument("--sets", type=int, default=4)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--only", type=int, default=-1, help="run only row i (0..7)")
    ap.add_argument("--lib", action="store_true", help="reference point: the same products (no epilogue) by torch.mm = the ROCm GEMM library")
    a = ap.parse_args()
    D, H = DIMS[a.model]
    M, I, R = 4 * D, H * 64, a.tokens
    dev, dt, td, f32 = "cuda:0", "bf16", torch.bfloat16, torch.float32
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, dtype=td: (torch.randn(*s, device=dev, generator=g) * 0.5).to(dtype)  # noqa: E731
    S = [dict(h=rn(R, D), qkv=rn(R, 3 * I), o=rn(R, I), u=rn(R, M), x32=rn(R, D, dtype=f32), wqkv=rn(3 * I, D), wqkv_t=rn(D, 3 * I),
              wo=rn(D, I), w1=rn(M, D), w1_t=rn(D, M), w2=rn(D, M), w2_t=rn(M, D), bD=rn(D, dtype=f32), bM=rn(M, dtype=f32),
              oq=torch.empty(R, 3 * I, device=dev, dtype=td), ou=torch.empty(R, M, device=dev, dtype=td),
              og=torch.empty(R, M, device=dev, dtype=td), oh=torch.empty(R, D, device=dev, dtype=td),
              ox=torch.empty(R, D, device=dev, dtype=f32)) for _ in range(a.sets)]
    rows = [
        ("to_qkv        (N=3I, K=D, store)", lambda s: ops.gemm_nt(s["h"], s["wqkv"], s["oq"], dt), 2.0 * R * 3 * I * D, R * (D + 3 * I) * 2),
        ("to_out + res  (N=D, K=I, fp32 out)", lambda s: ops.gemm_nt(s["o"], s["wo"], s["ox"], dt, epilogue=ops.EPI_BIAS_RES, bias=s["bD"], aux=s["x32"]),
         2.0 * R * D * I, R * (I * 2 + 8 * D)),
        ("net.0 + GELU  (N=M, K=D, 2 outs)", lambda s: ops.gemm_nt(s["h"], s["w1"], s["ou"], dt, epilogue=ops.EPI_BIAS_GELU, bias=s["bM"], out2=s["og"]),
         2.0 * R * M * D, R * (D + 2 * M) * 2),
        ("net.3 + res   (N=D, K=M, fp32 out)", lambda s: ops.gemm_nt(s["u"], s["w2"], s["ox"], dt, epilogue=ops.EPI_BIAS_RES, bias=s["bD"], aux=s["x32"]),
         2.0 * R * D * M, R * (M * 2 + 8 * D)),
        ("d net.3 gelu' (N=M, K=D, reads gd = gelu'(u) - 1/2)", lambda s: ops.gemm_nt(s["h"], s["w2_t"], s["ou"], dt, epilogue=ops.EPI_DGELU, aux=s["u"]),
         2.0 * R * D * M, R * (D + 2 * M) * 2),
        ("d net.0       (N=D, K=M, store)", lambda s: ops.gemm_nt(s["u"], s["w1_t"], s["oh"], dt), 2.0 * R * D * M, R * (M + D) * 2),
        ("d to_out      (N=I, K=D, store)", lambda s: ops.gemm_nt(s["h"], s["wo"].t().contiguous() if False else s["wo"], s["oh"], dt),
         2.0 * R * D * I, R * (D + I) * 2),
        ("d to_qkv      (N=D, K=3I, store)", lambda s: ops.gemm_nt(s["qkv"], s["wqkv_t"], s["oh"], dt), 2.0 * R * 3 * I * D, R * (3 * I + D) * 2),
    ]
    if a.lib:
        ops_lib = [("h", "wqkv", "oq"), ("o", "wo", "oh"), ("h", "w1", "ou"), ("u", "w2", "oh"), ("h", "w2_t", "ou"), ("u", "w1_t", "oh"),
                   ("h", "wo", "oh"), ("qkv", "wqkv_t", "oh")]
        rows = [(name + " [torch.mm]", (lambda s, k=k: torch.mm(s[k[0]], s[k[1]].t(), out=s[k[2]])), fl, nb)
                for (name, _, fl, nb), k in zip(rows, ops_lib)]
    if a.only >= 0:
        rows = rows[a.only:a.only + 1]
    tot = 0.0
    for name, fn, flops, nbytes in rows:
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
        tot += us
        print(f"{name:36s} {us:8.1f} us  {flops / us / 1e6:7.1f} TF/s ({flops / us / 1e6 / 25:.1f} %)  {nbytes / us / 1e3:7.1f} GB/s", flush=True)
    print(f"sum {tot:.1f} us")


if __name__ == "__main__":
    main()
