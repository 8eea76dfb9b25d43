"""Per-kernel timing of the hot path at the live workload's shapes (bench.py's `roofline` object).

Every kernel family of one encoder layer is launched exactly as csrc/encoder.hip launches it (same
entry point, operand dtypes, epilogue, workspace; the fused LayerNorm+MLP / LayerNorm+to_qkv kernels where
the shape supports them, the separate kernels otherwise) `reps` times inside a hipGraph, and the replay is
timed between two HIP events on the replay stream: no host launch gap is included.  The family with
the largest (average duration x launches per step) is the dominant kernel; its achieved rate =
algorithmic FLOPs (or bytes) per launch / average duration.  The rocprofv3 --kernel-trace --stats
summary of the same bench command (profiles/) lists the same kernels by name; an entry that covers two
kernels (e.g. the weight-gradient kernel and its slab reduction) must equal the sum of their averages.
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
    """[(name, rocprof kernel names, fn, flops, algorithmic bytes, launches per step)] for one layer."""
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
    x32, dx32, dxc = rn(R, D, dtype=f32), rn(R, D, dtype=f32), rn(R, D)
    wqkv, wqkv_t = rn(3 * I, D), rn(D, 3 * I)
    wo, wo_t = rn(D, I), rn(I, D)
    w1, w1_t, w2, w2_t = rn(M, D), rn(D, M), rn(D, M), rn(M, D)
    bD, bM = rn(D, dtype=f32), rn(M, dtype=f32)
    gam = rn(D, dtype=f32)
    out_qkv, out_o = torch.empty_like(qkv), torch.empty_like(o)
    out_x, out_xc = torch.empty_like(x32), torch.empty_like(h)
    out_u, out_g, out_h = torch.empty_like(u), torch.empty_like(u), torch.empty_like(h)
    dW = {k: torch.zeros(s, dtype=f32, device=dev) for k, s in
          dict(qkv=(3 * I, D), o=(D, I), w1=(M, D), w2=(D, M)).items()}
    dbD, dbD2, dbM = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(M, device=dev)
    o_att, lse = ops.attention_fwd(qkv, B, N, H, 0.125, dt)
    mean, rstd = torch.zeros(R, device=dev), torch.ones(R, device=dev)
    part = torch.empty(ops.layernorm_bwd_partial_floats(R, D), device=dev)
    probs = [dict(dY=dxc, X=gg, dW=dW["w2"], db=dbD), dict(dY=u, X=h, dW=dW["w1"], db=dbM),
             dict(dY=dxc, X=o, dW=dW["o"], db=dbD2), dict(dY=qkv, X=h, dW=dW["qkv"])]
    nbytes = ops.rt.lib.sitk_gemm_wgrad_group_ws_bytes(*_desc_array(probs), ops.rt.dtype_code(dt))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    att = 4.0 * B * H * N * N * 64
    wg_flops = 2.0 * R * (D * M * 2 + D * I + 3 * I * D)
    mlp_flops = 4.0 * R * D * M
    fused_mlp = ops.mlp_fused_supported(D, M, dt)
    fused_qkv = ops.ln_gemm_fused_supported(D, 3 * I, dt)
    rows = []
    # ---- forward ----
    fused_chain = fused_mlp and fused_qkv and ops.attn_out_mlp_fused_supported(R, D, I, M, dt) and L > 1
    if fused_qkv:
        rows.append(("norm + to_qkv (fused)", "ln_gemm_fwd_kernel", lambda: ops.ln_gemm_fwd(x32, gam, bD, wqkv, dt),
                     2.0 * R * 3 * I * D, R * (4 * D + (D + 3 * I) * es), 1 if fused_chain else L))
    else:
        rows.append(("layernorm_fwd", "layernorm_fwd_kernel", lambda: ops.layernorm_fwd(x32, gam, bD, dt), 0, R * D * (4 + es), L))
        rows.append(("to_qkv", "gemm_nt_wres_kernel", lambda: ops.gemm_nt(h, wqkv, out_qkv, dt), 2.0 * R * 3 * I * D, R * (D + 3 * I) * es, L))
    rows.append(("attention forward", "attn_fwd_res_kernel", lambda: ops.attention_fwd(qkv, B, N, H, 0.125, dt), att, R * 4 * I * es, L))
    fused_proj = fused_mlp and ops.attn_out_mlp_fused_supported(R, D, I, M, dt)
    if fused_proj and fused_qkv and L > 1:
        # blocks 0 .. L-2: one launch from the attention output to the next block's qkv; the last block stops at `out`
        rows.append(("to_out + norm + MLP + residuals + next block's norm + to_qkv (fused)", "mlp_kernel<false, .., NEXT>",
                     lambda: ops.attn_out_mlp_next_fwd(o, wo, bD, x32, gam, bD, w1, bM, w2, bD, gam, bD, wqkv, dt, want_g=True),
                     mlp_flops + 2.0 * R * D * I + 2.0 * R * 3 * I * D, R * (I * es + 12 * D + (2 * D + 2 * M + 3 * I) * es), L - 1))
        rows.append(("to_out + residual + norm + net.0 + GELU + net.3 + residual (fused, last block)", "mlp_kernel<false>",
                     lambda: ops.attn_out_mlp_fwd(o, wo, bD, x32, gam, bD, w1, bM, w2, bD, dt, want_g=True),
                     mlp_flops + 2.0 * R * D * I, R * (I * es + 12 * D + (D + 2 * M) * es), 1))
    elif fused_proj:
        rows.append(("to_out + residual + norm + net.0 + GELU + net.3 + residual (fused)", "mlp_kernel<false>",
                     lambda: ops.attn_out_mlp_fwd(o, wo, bD, x32, gam, bD, w1, bM, w2, bD, dt, want_g=True),
                     mlp_flops + 2.0 * R * D * I, R * (I * es + 12 * D + (D + 2 * M) * es), L))
    else:
        rows.append(("to_out + residual", "gemm_nt_wres_kernel",
                     lambda: ops.gemm_nt(o, wo, out_x, dt, epilogue=ops.EPI_BIAS_RES, bias=bD, aux=x32), 2.0 * R * D * I, R * (I * es + 8 * D), L))
    if fused_proj:
        pass
    elif fused_mlp:
        rows.append(("norm + net.0 + GELU + net.3 + residual (fused)", "mlp_kernel<false>",
                     lambda: ops.mlp_fwd(x32, gam, bD, w1, bM, w2, bD, dt, want_g=True), mlp_flops,
                     R * (8 * D + (D + 2 * M) * es), L))
    else:
        rows.append(("layernorm_fwd (2)", "layernorm_fwd_kernel", lambda: ops.layernorm_fwd(x32, gam, bD, dt), 0, R * D * (4 + es), L))
        rows.append(("net.0 + GELU", "gemm_nt_wres_kernel",
                     lambda: ops.gemm_nt(h, w1, out_u, dt, epilogue=ops.EPI_BIAS_GELU, bias=bM, out2=out_g), 2.0 * R * M * D, R * (D + 2 * M) * es, L))
        rows.append(("net.3 + residual", "gemm_nt_n192_kernel",
                     lambda: ops.gemm_nt(gg, w2, out_x, dt, epilogue=ops.EPI_BIAS_RES, bias=bD, aux=x32), 2.0 * R * D * M, R * (M * es + 8 * D), L))
    # ---- backward ----
    if fused_mlp:
        rows.append(("d net.3 x GELU' + d net.0 + norm backward (fused)", "mlp_kernel<true>",
                     lambda: ops.mlp_bwd(dx32, dxc, x32, mean, rstd, gam, w2_t, w1_t, u, dt, want_g=False), mlp_flops,
                     R * (D * es + 2 * M * es + 12 * D + D * es), L))
    else:
        rows.append(("d net.3 (x GELU')", "gemm_nt_wres_kernel",
                     lambda: ops.gemm_nt(dxc, w2_t, out_u, dt, epilogue=ops.EPI_DGELU, aux=u), 2.0 * R * D * M, R * (D + 2 * M) * es, L))
        rows.append(("d net.0", "gemm_nt_n192_kernel", lambda: ops.gemm_nt(u, w1_t, out_h, dt), 2.0 * R * D * M, R * (M + D) * es, L))
        rows.append(("layernorm_bwd (2)", "layernorm_bwd_kernel",
                     lambda: ops.layernorm_bwd(h, x32, mean, rstd, gam, dx32, dbD, dbD2, dt, dx=out_x, dx_c=out_xc, partials=part),
                     0, R * D * (2 * es + 12), L))
    if ops.attention_bwd_proj_supported(N, D, dt):     # d to_out folded into the query-side kernel (csrc/encoder.hip)
        rows.append(("d to_out + attention backward (fused)", "attn_bwd_dq_res_kernel + attn_bwd_dkv_res_kernel",
                     lambda: ops.attention_bwd_proj(qkv, o_att, dxc, wo_t, lse, B, N, H, 0.125, dt),
                     2.5 * att + 2.0 * R * D * I, R * (8 * I + D) * es, L))
    else:
        rows.append(("d to_out", "gemm_nt_wres_kernel", lambda: ops.gemm_nt(dxc, wo_t, out_o, dt), 2.0 * R * D * I, R * (D + I) * es, L))
        rows.append(("attention backward", "attn_bwd_dq_res_kernel + attn_bwd_dkv_res_kernel",
                     lambda: ops.attention_bwd(qkv, o_att, o, lse, B, N, H, 0.125, dt), 2.5 * att, R * 8 * I * es, L))
    # the weight gradients of a whole backward slice run as one launch (csrc/encoder.hip): all L layers on one GPU
    if 4 * L <= 48 and nbytes > 0:      # distinct operand tensors per layer, as in the real step (1.4 GB for tiny)
        probs_all = list(probs)
        for _ in range(L - 1):
            u_l, gg_l, h_l, o_l, qkv_l, dxc_l = (t.clone() for t in (u, gg, h, o, qkv, dxc))
            probs_all += [dict(dY=dxc_l, X=gg_l, dW=dW["w2"], db=dbD), dict(dY=u_l, X=h_l, dW=dW["w1"], db=dbM),
                          dict(dY=dxc_l, X=o_l, dW=dW["o"], db=dbD2), dict(dY=qkv_l, X=h_l, dW=dW["qkv"])]
    else:
        probs_all = probs
    nl = len(probs_all) // 4
    nb_all = ops.rt.lib.sitk_gemm_wgrad_group_ws_bytes(*_desc_array(probs_all), ops.rt.dtype_code(dt))
    ws_all = torch.empty(max(nb_all, nbytes, 16), dtype=torch.uint8, device=dev)
    rows.append((f"weight gradients of {nl} layer(s), one launch", "wgrad_big_kernel + wgrad_big_reduce_kernel",
                 lambda: ops.gemm_wgrad_group(probs_all, dt, workspace=ws_all), wg_flops * nl,
                 nl * R * (2 * D + 2 * M + 4 * I + 2 * D) * es, L // nl))
    if fused_qkv:
        rows.append(("d to_qkv + norm backward (fused)", "ln_gemm_bwd_kernel",
                     lambda: ops.ln_gemm_bwd(qkv, wqkv_t, x32, mean, rstd, gam, dx32, dt), 2.0 * R * 3 * I * D,
                     R * (3 * I * es + 12 * D + D * es), L))
    else:
        rows.append(("d to_qkv", "gemm_nt_n192_kernel", lambda: ops.gemm_nt(qkv, wqkv_t, out_h, dt), 2.0 * R * 3 * I * D, R * (3 * I + D) * es, L))
        rows.append(("layernorm_bwd", "layernorm_bwd_kernel",
                     lambda: ops.layernorm_bwd(h, x32, mean, rstd, gam, dx32, dbD, dbD2, dt, dx=out_x, dx_c=out_xc, partials=part),
                     0, R * D * (2 * es + 12), L))
    return rows


def _desc_array(problems):
    from . import runtime as rt
    arr = (rt.WgradDesc * len(problems))()
    for d, p in zip(arr, problems):
        dY, X, dW = p["dY"], p["X"], p["dW"]
        d.M, d.N, d.K = dY.shape[0], dW.shape[0], dW.shape[1]
        d.lddy, d.ldx, d.lddw = dY.stride(0), X.stride(0), dW.stride(0)
        d.dy_is_f32 = int(dY.dtype == torch.float32)
    return arr, len(problems)


def dominant_kernel_roofline(eng, peak_tflops, peak_gbs, reps=20):
    rows = []
    for name, kernels, fn, flops, nbytes, launches in layer_kernels(eng):
        t = _time(fn, reps)
        rows.append(dict(op=name, kernels=kernels, us=round(t * 1e6, 2), launches_per_step=launches,
                         tflops=round(flops / t / 1e12, 1), gbs=round(nbytes / t / 1e9, 1),
                         step_share_us=round(t * 1e6 * launches, 1)))
    dom = max(rows, key=lambda r: r["step_share_us"])
    total = sum(r["step_share_us"] for r in rows)
    mfma_bound = dom["tflops"] > 0
    ach = dom["tflops"] if mfma_bound else dom["gbs"]
    peak = peak_tflops if mfma_bound else peak_gbs
    return {"bound": "mfma" if mfma_bound else "hbm", "kernel": dom["kernels"], "op": dom["op"], "achieved": ach,
            "peak": peak, "unit": "TFLOP/s" if mfma_bound else "GB/s", "frac": round(ach / peak, 4), "traffic": None,
            "avg_us": dom["us"], "launches_per_step": dom["launches_per_step"],
            "encoder_kernel_sum_us": round(total, 1), "kernels": rows}
