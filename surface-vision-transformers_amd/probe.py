"""Per-kernel timing of the hot path at the live workload's shapes (bench.py's `roofline` object).

Every kernel family of one encoder layer is launched `reps` times back to back on torch's current
stream between two HIP events (the same stream the engine launches on), on buffers of exactly the
shapes the engine uses.  The family with the largest (average duration x launches per step) is the
dominant kernel; its achieved rate = algorithmic FLOPs (or bytes) per launch / average duration.
The rocprofv3 --kernel-trace --stats summary of the same bench command (profiles/) lists the same
kernels by name; their average durations must agree with these.
"""
import torch

from . import ops


def _time(fn, reps):
    """Average device time of one launch: `reps` launches captured in a hipGraph (so no host launch
    gap is timed), replayed between two HIP events on the replay stream."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3   # seconds


def layer_kernels(eng):
    """[(name, fn, flops, algorithmic bytes, launches per step)] for one encoder layer."""
    B, N, D, dt = eng.B, eng.N, eng.D, eng.dtype
    tr = eng.sit.transformer
    H, M = tr.heads, tr.mlp_dim
    I, R = H * 64, B * N
    dev, td = eng.device, eng.tdt
    es = 2 if td == torch.bfloat16 else 4
    L = eng.depth
    g = torch.Generator(device=dev).manual_seed(7)
    rn = lambda *s, dtype=td: (torch.randn(*s, device=dev, generator=g) * 0.5).to(dtype)  # noqa: E731
    f32 = torch.float32
    h, qkv, o, u, gg = rn(R, D), rn(R, 3 * I), rn(R, I), rn(R, M), rn(R, M)
    x32, dx32 = rn(R, D, dtype=f32), rn(R, D, dtype=f32)
    wqkv, wqkv_t = rn(3 * I, D), rn(D, 3 * I)
    wo, wo_t = rn(D, I), rn(I, D)
    w1, w1_t, w2, w2_t = rn(M, D), rn(D, M), rn(D, M), rn(M, D)
    bD, bM = rn(D, dtype=f32), rn(M, dtype=f32)
    gam = rn(D, dtype=f32)
    out_qkv, out_o = torch.empty_like(qkv), torch.empty_like(o)
    out_x = torch.empty_like(x32)
    out_u, out_g, out_h = torch.empty_like(u), torch.empty_like(u), torch.empty_like(h)
    dW = {k: torch.zeros(s, dtype=f32, device=dev) for k, s in
          dict(qkv=(3 * I, D), o=(D, I), w1=(M, D), w2=(D, M)).items()}
    dbD, dbM = torch.zeros(D, device=dev), torch.zeros(M, device=dev)
    o_att, lse = ops.attention_fwd(qkv, B, N, H, 0.125, dt)
    mean, rstd = torch.zeros(R, device=dev), torch.ones(R, device=dev)
    att = 4.0 * B * H * N * N * 64
    ks = [
        ("layernorm_fwd", lambda: ops.layernorm_fwd(x32, gam, bD, dt), 0, R * D * (4 + es), 2 * L),
        ("gemm_nt qkv (STORE)", lambda: ops.gemm_nt(h, wqkv, out_qkv, dt), 2.0 * R * 3 * I * D, R * (D + 3 * I) * es, L),
        ("attn_fwd_kernel", lambda: ops.attention_fwd(qkv, B, N, H, 0.125, dt), att, R * 4 * I * es, L),
        ("gemm_nt proj (BIAS_RES)", lambda: ops.gemm_nt(o, wo, out_x, dt, epilogue=ops.EPI_BIAS_RES, bias=bD, aux=x32),
         2.0 * R * D * I, R * (I * es + 8 * D), L),
        ("gemm_nt fc1 (BIAS_GELU)", lambda: ops.gemm_nt(h, w1, out_u, dt, epilogue=ops.EPI_BIAS_GELU, bias=bM, out2=out_g),
         2.0 * R * M * D, R * (D + 2 * M) * es, L),
        ("gemm_nt fc2 (BIAS_RES)", lambda: ops.gemm_nt(gg, w2, out_x, dt, epilogue=ops.EPI_BIAS_RES, bias=bD, aux=x32),
         2.0 * R * D * M, R * (M * es + 8 * D), L),
        ("wgrad w2", lambda: ops.gemm_wgrad(dx32, gg, dW["w2"], dt, db=dbD), 2.0 * R * D * M, R * (4 * D + M * es), L),
        ("gemm_nt dfc2 (DGELU)", lambda: ops.gemm_nt(dx32, w2_t, out_u, dt, epilogue=ops.EPI_DGELU, aux=u),
         2.0 * R * D * M, R * (4 * D + 2 * M * es), L),
        ("wgrad w1", lambda: ops.gemm_wgrad(u, h, dW["w1"], dt, db=dbM), 2.0 * R * D * M, R * (M + D) * es, L),
        ("gemm_nt dfc1 (STORE)", lambda: ops.gemm_nt(u, w1_t, out_h, dt), 2.0 * R * D * M, R * (M + D) * es, L),
        ("layernorm_bwd", lambda: ops.layernorm_bwd(h, x32, mean, rstd, gam, dx32, dbD, dbD, dt, dx=out_x), 0,
         R * D * (es + 12), 2 * L),
        ("wgrad wo", lambda: ops.gemm_wgrad(dx32, o, dW["o"], dt, db=dbD), 2.0 * R * D * I, R * (4 * D + I * es), L),
        ("gemm_nt dproj (STORE)", lambda: ops.gemm_nt(dx32, wo_t, out_o, dt), 2.0 * R * D * I, R * (4 * D + I * es), L),
        ("attention_bwd (dq + dkv kernels)", lambda: ops.attention_bwd(qkv, o_att, o, lse, B, N, H, 0.125, dt),
         2.5 * att, R * 8 * I * es, L),
        ("wgrad wqkv", lambda: ops.gemm_wgrad(qkv, h, dW["qkv"], dt), 2.0 * R * 3 * I * D, R * (3 * I + D) * es, L),
        ("gemm_nt dqkv (STORE)", lambda: ops.gemm_nt(qkv, wqkv_t, out_h, dt), 2.0 * R * 3 * I * D, R * (3 * I + D) * es, L),
    ]
    return ks


def dominant_kernel_roofline(eng, peak_tflops, peak_gbs, reps=20):
    rows = []
    for name, fn, flops, nbytes, launches in layer_kernels(eng):
        t = _time(fn, reps)
        rows.append(dict(kernel=name, us=round(t * 1e6, 2), launches_per_step=launches,
                         tflops=round(flops / t / 1e12, 1), gbs=round(nbytes / t / 1e9, 1),
                         step_share_us=round(t * 1e6 * launches, 1)))
    dom = max(rows, key=lambda r: r["step_share_us"])
    total = sum(r["step_share_us"] for r in rows)
    mfma_bound = dom["tflops"] > 0
    ach = dom["tflops"] if mfma_bound else dom["gbs"]
    peak = peak_tflops if mfma_bound else peak_gbs
    return {"bound": "mfma" if mfma_bound else "hbm", "kernel": dom["kernel"], "achieved": ach, "peak": peak,
            "unit": "TFLOP/s" if mfma_bound else "GB/s", "frac": round(ach / peak, 4), "traffic": None,
            "avg_us": dom["us"], "launches_per_step": dom["launches_per_step"],
            "encoder_kernel_sum_us": round(total, 1), "kernels": rows}
