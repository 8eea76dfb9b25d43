"""Per-kernel timing of the hot path at the live workload's shapes (bench.py's `roofline` object).

Every kernel of one encoder layer is launched exactly as csrc/encoder.hip launches it (same entry point, operand
dtypes, epilogue, workspace; the fused kernels where the shape supports them, the separate ones otherwise) `reps` times
inside a hipGraph, and the replay is timed between two HIP events on the replay stream: no host launch gap is included.

Each repetition works on its OWN set of operand and output buffers (one set per layer, as in the real step, where
every layer's activations are distinct tensors): replaying one launch over the same buffers lets its stores hit lines
the previous repetition left in the Infinity Cache and reads 20-25 % fast (round 1's table did that).  Rows are single
kernels (round 4: the attention backward is ONE launch where the sequence is LDS-resident, and d to_qkv + norm backward of
layer l runs in one launch with the MLP backward of layer l - 1), so each
`us` must agree with the same kernel's average in profiles/*step_kernel_stats*.md of the same bench command.

The dominant kernel is the single kernel with the largest (average duration x launches per step).  Its bound is the
LARGER of its two roofline fractions: algorithmic FLOPs / duration against the dense bf16 MFMA peak, algorithmic bytes
/ duration against the HBM peak.
"""
import torch

from . import ops
from . import runtime as rt


def _time(fn, nset, reps):
    """Average device time of one launch: `reps` launches (launch i on buffer set i % nset) captured in a hipGraph,
    replayed between two HIP events on the replay stream.  Outputs stay alive through the capture so that no two
    launches share an output buffer through the caching allocator."""
    for i in range(min(2, nset)):
        fn(i)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    keep = []
    with torch.cuda.graph(graph):
        for i in range(reps):
            keep.append(fn(i % nset))
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    e1.synchronize()
    dt = e0.elapsed_time(e1) / reps * 1e-3   # seconds
    del keep, graph
    return dt


class _Set:
    """One layer's worth of operands (random, scaled like activations)."""

    def __init__(self, eng, seed):
        B, N, D = eng.B, eng.N, eng.D
        tr = eng.sit.transformer
        H, M = tr.heads, tr.mlp_dim
        I, R = H * 64, B * N
        dev, td, f32 = eng.device, eng.tdt, torch.float32
        g = torch.Generator(device=dev).manual_seed(seed)
        rn = lambda *s, dtype=td: (torch.randn(*s, device=dev, generator=g) * 0.5).to(dtype)  # noqa: E731
        self.h, self.qkv, self.o, self.u, self.gg = rn(R, D), rn(R, 3 * I), rn(R, I), rn(R, M), rn(R, M)
        self.x32, self.dx32, self.dxc = rn(R, D, dtype=f32), rn(R, D, dtype=f32), rn(R, D)
        self.wqkv, self.wqkv_t = rn(3 * I, D), rn(D, 3 * I)
        self.wo, self.wo_t = rn(D, I), rn(I, D)
        self.w1, self.w1_t, self.w2, self.w2_t = rn(M, D), rn(D, M), rn(D, M), rn(M, D)
        self.bD, self.bM, self.gam = rn(D, dtype=f32), rn(M, dtype=f32), rn(D, dtype=f32)
        self.mean, self.rstd = torch.zeros(R, device=dev), torch.ones(R, device=dev)
        self.o_att, self.lse = ops.attention_fwd(self.qkv, B, N, H, 0.125, eng.dtype)
        self.delta = torch.zeros_like(self.lse)
        self.d_o = rn(R, I)


def layer_kernels(eng, nset):
    """[(op, rocprof kernel name, fn(set index), flops, algorithmic bytes, launches per step)] for one layer."""
    B, N, D, dt = eng.B, eng.N, eng.D, eng.dtype
    tr = eng.sit.transformer
    H, M = tr.heads, tr.mlp_dim
    I, R = H * 64, B * N
    dev, td = eng.device, eng.tdt
    es = 4 if td == torch.float32 else 2
    L = eng.depth
    f32 = torch.float32
    S = [_Set(eng, 7 + i) for i in range(nset)]
    dW = {k: torch.zeros(s, dtype=f32, device=dev) for k, s in
          dict(qkv=(3 * I, D), o=(D, I), w1=(M, D), w2=(D, M)).items()}
    dbD, dbD2, dbM = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(M, device=dev)
    part = torch.empty(ops.layernorm_bwd_partial_floats(R, D), device=dev)
    att = 4.0 * B * H * N * N * 64
    wg_flops = 2.0 * R * (D * M * 2 + D * I + 3 * I * D)
    mlp_flops = 4.0 * R * D * M
    fused_mlp = ops.mlp_fused_supported(D, M, dt)
    fused_qkv = ops.ln_gemm_fused_supported(D, 3 * I, dt)
    fused_proj = fused_mlp and ops.attn_out_mlp_fused_supported(R, D, I, M, dt)
    fused_chain = fused_proj and fused_qkv and L > 1
    rows = []
    # op name -> label of the timeline marks csrc/encoder.hip records behind that launch
    TL = {"norm + to_qkv (fused)": "ln_gemm_fwd", "norm (attention)": "layernorm_fwd", "to_qkv": "gemm:to_qkv", "attention forward": "attn_fwd", "to_out + norm + MLP + residuals + next block's norm + to_qkv (fused)": "block_tail_next", "to_out + residual + norm + net.0 + GELU + net.3 + residual (fused)": "block_tail", "to_out + residual": "gemm:to_out", "norm + net.0 + GELU + net.3 + residual (fused)": "mlp_fwd", "norm (MLP)": "layernorm_fwd", "net.0 + GELU": "gemm:net0", "net.3 + residual": "gemm:net3", "d net.3 x GELU' + d net.0 + norm backward (fused)": "mlp_bwd", "d net.3 (x GELU')": "gemm:dnet3", "d net.0": "gemm:dnet0", "norm backward (MLP)": "layernorm_bwd", "d to_out": "gemm:dto_out", "d to_qkv + norm backward (fused)": "ln_gemm_bwd", "d to_qkv": "gemm:dqkv", "norm backward (attention)": "layernorm_bwd"}

    def add(name, kernel, fn, flops, nbytes, launches, label=None):
        rows.append((name, kernel, fn, flops, nbytes, launches, label or TL.get(name)))

    def ln_bwd(i):
        dxc = torch.empty_like(S[i].h)
        return dxc, ops.layernorm_bwd(S[i].h, S[i].x32, S[i].mean, S[i].rstd, S[i].gam, S[i].dx32, dbD, dbD2, dt,
                                      dx=torch.empty_like(S[i].x32), dx_c=dxc, partials=part)
    # ---- forward ----
    if fused_qkv:
        add("norm + to_qkv (fused)", "ln_gemm_fwd_kernel", lambda i: ops.ln_gemm_fwd(S[i].x32, S[i].gam, S[i].bD, S[i].wqkv, dt),
            2.0 * R * 3 * I * D, R * (4 * D + (D + 3 * I) * es), 1 if fused_chain else L)
    else:
        add("norm (attention)", "layernorm_fwd_kernel", lambda i: ops.layernorm_fwd(S[i].x32, S[i].gam, S[i].bD, dt), 0, R * D * (4 + es), L)
        add("to_qkv", "gemm_nt", lambda i: ops.gemm_nt(S[i].h, S[i].wqkv, torch.empty_like(S[i].qkv), dt),
            2.0 * R * 3 * I * D, R * (D + 3 * I) * es, L)
    add("attention forward", "attn_fwd", lambda i: ops.attention_fwd(S[i].qkv, B, N, H, 0.125, dt), att, R * 4 * I * es, L)
    if fused_chain:
        # blocks 0 .. L-2: one launch from the attention output to the next block's qkv; the last block stops at `out`
        add("to_out + norm + MLP + residuals + next block's norm + to_qkv (fused)", "mlp_kernel<false, .., NEXT>",
            lambda i: ops.attn_out_mlp_next_fwd(S[i].o, S[i].wo, S[i].bD, S[i].x32, S[i].gam, S[i].bD, S[i].w1, S[i].bM, S[i].w2,
                                                S[i].bD, S[i].gam, S[i].bD, S[i].wqkv, dt, want_g=True),
            mlp_flops + 2.0 * R * D * I + 2.0 * R * 3 * I * D, R * (I * es + 12 * D + (2 * D + 2 * M + 3 * I) * es), L - 1)
    if fused_proj:
        add("to_out + residual + norm + net.0 + GELU + net.3 + residual (fused)", "mlp_kernel<false, .., PROJ>",
            lambda i: ops.attn_out_mlp_fwd(S[i].o, S[i].wo, S[i].bD, S[i].x32, S[i].gam, S[i].bD, S[i].w1, S[i].bM, S[i].w2, S[i].bD,
                                           dt, want_g=True),
            mlp_flops + 2.0 * R * D * I, R * (I * es + 12 * D + (D + 2 * M) * es), 1 if fused_chain else L)
    else:
        add("to_out + residual", "gemm_nt", lambda i: ops.gemm_nt(S[i].o, S[i].wo, torch.empty_like(S[i].x32), dt,
                                                                   epilogue=ops.EPI_BIAS_RES, bias=S[i].bD, aux=S[i].x32),
            2.0 * R * D * I, R * (I * es + 8 * D), L)
        if fused_mlp:
            add("norm + net.0 + GELU + net.3 + residual (fused)", "mlp_kernel<false>",
                lambda i: ops.mlp_fwd(S[i].x32, S[i].gam, S[i].bD, S[i].w1, S[i].bM, S[i].w2, S[i].bD, dt, want_g=True), mlp_flops,
                R * (8 * D + (D + 2 * M) * es), L)
        else:
            add("norm (MLP)", "layernorm_fwd_kernel", lambda i: ops.layernorm_fwd(S[i].x32, S[i].gam, S[i].bD, dt), 0, R * D * (4 + es), L)
            def fc1(i):
                o2 = torch.empty_like(S[i].u)
                return o2, ops.gemm_nt(S[i].h, S[i].w1, torch.empty_like(S[i].u), dt, epilogue=ops.EPI_BIAS_GELU, bias=S[i].bM, out2=o2)
            add("net.0 + GELU", "gemm_nt", fc1, 2.0 * R * M * D, R * (D + 2 * M) * es, L)
            add("net.3 + residual", "gemm_nt", lambda i: ops.gemm_nt(S[i].gg, S[i].w2, torch.empty_like(S[i].x32), dt,
                                                                      epilogue=ops.EPI_BIAS_RES, bias=S[i].bD, aux=S[i].x32),
                2.0 * R * D * M, R * (M * es + 8 * D), L)
    # ---- backward ----
    if fused_mlp:
        add("d net.3 x GELU' + d net.0 + norm backward (fused)", "mlp_kernel<true>",
            lambda i: ops.mlp_bwd(S[i].dx32, S[i].dxc, S[i].x32, S[i].mean, S[i].rstd, S[i].gam, S[i].w2_t, S[i].w1_t, S[i].u, dt),
            mlp_flops, R * (D * es + 2 * M * es + 12 * D + D * es), L)
    else:
        add("d net.3 (x GELU')", "gemm_nt", lambda i: ops.gemm_nt(S[i].dxc, S[i].w2_t, torch.empty_like(S[i].u), dt, epilogue=ops.EPI_DGELU,
                                                                   aux=S[i].u), 2.0 * R * D * M, R * (D + 2 * M) * es, L)
        add("d net.0", "gemm_nt", lambda i: ops.gemm_nt(S[i].u, S[i].w1_t, torch.empty_like(S[i].h), dt), 2.0 * R * D * M, R * (M + D) * es, L)
        add("norm backward (MLP)", "layernorm_bwd_kernel", ln_bwd, 0, R * D * (2 * es + 12), L)
    fold = ops.attention_bwd_proj_supported(N, D, dt)     # d to_out folded into the query-side kernel (csrc/encoder.hip)
    if not fold:
        add("d to_out", "gemm_nt", lambda i: ops.gemm_nt(S[i].dxc, S[i].wo_t, torch.empty_like(S[i].o), dt), 2.0 * R * D * I, R * (D + I) * es, L)

    def att_bwd(i, phases):
        s = S[i]
        dqkv = torch.empty_like(s.qkv)
        rt.check(rt.lib.sitk_attention_bwd_phases(
            s.qkv.data_ptr(), s.o_att.data_ptr(), None if fold else s.d_o.data_ptr(), s.dxc.data_ptr() if fold else None,
            s.wo_t.data_ptr() if fold else None, s.d_o.data_ptr() if fold else None, s.lse.data_ptr(), s.delta.data_ptr(),
            dqkv.data_ptr(), B, N, H, D, 0.125, dt, phases, rt.stream_ptr()))
        return dqkv
    # algorithmic work (backward = 2 x forward = four products: dP, dQ, dV, dK; both sides also recompute S, the
    # query side executes the folded projection).  ONE launch where the sequence is LDS-resident (round 4: query side + key
    # side merged, q / k / v cross HBM once), two launches timed as one row otherwise (one timeline mark behind the pair).
    add("attention backward (dQ" + (", d to_out folded in" if fold else "") + ", dK, dV)", "attn_bwd",
        lambda i: att_bwd(i, 3), 2 * att + (2.0 * R * D * I if fold else 0.0), R * (3 * I + I + (D + I if fold else I) + 3 * I) * es, L,
        "attn_bwd")
    # the weight gradients of a whole backward slice run as one launch (csrc/encoder.hip): all L layers on one GPU
    probs = []
    nl = L if 4 * L <= 48 else 1
    for l in range(nl):        # distinct operand tensors per layer, as in the real step
        s = S[l % nset]
        u_l, gg_l, h_l, o_l, qkv_l, dxc_l = (t if l < nset else t.clone() for t in (s.u, s.gg, s.h, s.o, s.qkv, s.dxc))
        probs += [dict(dY=dxc_l, X=gg_l, dW=dW["w2"], db=dbD), dict(dY=u_l, X=h_l, dW=dW["w1"], db=dbM),
                  dict(dY=dxc_l, X=o_l, dW=dW["o"], db=dbD2), dict(dY=qkv_l, X=h_l, dW=dW["qkv"])]
    nb_all = rt.lib.sitk_gemm_wgrad_group_ws_bytes(*_desc_array(probs), rt.dtype_code(dt))
    if nb_all == 0 and nl > 1:          # shapes outside the batched large-tile path: one layer per launch
        probs, nl = probs[:4], 1
        nb_all = rt.lib.sitk_gemm_wgrad_group_ws_bytes(*_desc_array(probs), rt.dtype_code(dt))
    ws_all = torch.empty(max(nb_all, 16), dtype=torch.uint8, device=dev)
    add(f"weight gradients of {nl} layer(s), one launch", "wgrad_x2_kernel" if nb_all else "wgrad_kernel",
        lambda i: ops.gemm_wgrad_group(probs, dt, workspace=ws_all if nb_all else None), wg_flops * nl,
        nl * R * (2 * D + 2 * M + 4 * I + 2 * D) * es, L // nl, "wgrad")
    pair = fused_qkv and fused_mlp and L > 1 and ops.ln_gemm_mlp_bwd_supported(R, D, 3 * I, M, dt)
    if pair:
        # layers L-1 .. 1: d to_qkv + norm backward of layer l and the MLP backward of layer l - 1 in ONE launch (round 4); the
        # first MLP backward and the last d to_qkv of a slice stay single launches (the two rows around this one)
        rows[:] = [(n, k, f, fl, nb, (1 if lab == "mlp_bwd" else ln), lab) for (n, k, f, fl, nb, ln, lab) in rows]
        add("d to_qkv + norm backward + next layer's d net.3 x GELU' + d net.0 + norm backward (one launch)", "ln_gemm_mlp_bwd_kernel",
            lambda i: ops.ln_gemm_mlp_bwd(S[i].qkv, S[i].wqkv_t, S[i].x32, S[i].mean, S[i].rstd, S[i].gam, S[i].dx32, S[i].x32,
                                          S[i].mean, S[i].rstd, S[i].gam, S[i].w2_t, S[i].w1_t, S[i].u, dt),
            # bytes that have to cross HBM: the two kernels' sums minus dx (fp32) and its compute-dtype copy, which the second
            # half reads back from L2 (its own workgroup wrote them)
            2.0 * R * 3 * I * D + mlp_flops, R * (3 * I * es + 12 * D + D * es) + R * (2 * M * es + 8 * D + D * es), L - 1,
            "ln_gemm_mlp_bwd")
    if fused_qkv:
        add("d to_qkv + norm backward (fused)", "ln_gemm_bwd_kernel",
            lambda i: ops.ln_gemm_bwd(S[i].qkv, S[i].wqkv_t, S[i].x32, S[i].mean, S[i].rstd, S[i].gam, S[i].dx32, dt),
            2.0 * R * 3 * I * D, R * (3 * I * es + 12 * D + D * es), 1 if pair else L)
    else:
        add("d to_qkv", "gemm_nt", lambda i: ops.gemm_nt(S[i].qkv, S[i].wqkv_t, torch.empty_like(S[i].h), dt), 2.0 * R * 3 * I * D,
            R * (3 * I + D) * es, L)
        add("norm backward (attention)", "layernorm_bwd_kernel", ln_bwd, 0, R * D * (2 * es + 12), L)
    return rows


def _desc_array(problems):
    arr = (rt.WgradDesc * len(problems))()
    for d, p in zip(arr, problems):
        dY, X, dW = p["dY"], p["X"], p["dW"]
        d.M, d.N, d.K = dY.shape[0], dW.shape[0], dW.shape[1]
        d.lddy, d.ldx, d.lddw = dY.stride(0), X.stride(0), dW.stride(0)
        d.dy_is_f32 = int(dY.dtype == torch.float32)
    return arr, len(problems)


def step_timeline(eng, reps=3):
    """Per-kernel device times INSIDE the real step: one train step runs eagerly with a timeline attached to the encoder
    (csrc/encoder.hip records a HIP event behind every launch); returns {label: (average us, launches per step)}.  An
    interval runs from the previous kernel's end to this kernel's end, i.e. it includes the ~1-2 us the hardware needs to
    start a dependent kernel."""
    import ctypes as C
    if eng.dp:
        return {}
    cap = 4096
    tl = rt.lib.sitk_timeline_create(cap)
    if not tl:
        return {}
    us = (C.c_float * cap)()
    lab = (C.c_char_p * cap)()
    acc = {}
    try:
        eng.cfg.timeline = tl
        for r in range(reps + 1):
            rt.lib.sitk_timeline_reset(tl)
            for fn in eng._segment_fns():
                fn()
            eng._finish_backward()
            eng._optimizer()
            n = rt.lib.sitk_timeline_read(tl, us, lab, cap)
            if r == 0 or n <= 0:
                continue                                  # first pass: warm-up
            for i in range(n):
                k = lab[i].decode()
                if k == "begin":
                    continue                              # the gap between forward and backward (head, loss) is not a kernel
                a = acc.setdefault(k, [0.0, 0])
                a[0] += us[i]
                a[1] += 1
    finally:
        eng.cfg.timeline = None
        rt.lib.sitk_timeline_destroy(tl)
    return {k: (v[0] / v[1], v[1] // reps) for k, v in acc.items()}


def dominant_kernel_roofline(eng, peak_tflops, peak_gbs, reps=None):
    """`us` of every row = the kernel's average time inside the real step (step_timeline; what rocprofv3 --kernel-trace
    --stats of the same command reports, plus the dependent-launch gap); `us_isolated` = the same launch replayed from a
    hipGraph over rotating cold buffers.  The dominant kernel = the single kernel with the largest in-step share."""
    tr = eng.sit.transformer
    R, D, M, I = eng.B * eng.N, eng.D, tr.mlp_dim, tr.heads * 64
    per_set = R * (2 * (2 * D + 4 * I + 2 * M + D + I) + 8 * D) * 2.2     # operands + the outputs kept alive, bytes
    nset = int(max(2, min(eng.depth, 12, 40e9 // per_set)))
    reps = reps or nset
    tline = step_timeline(eng)
    rows = []
    for name, kernel, fn, flops, nbytes, launches, label in layer_kernels(eng, nset):
        t_iso = _time(fn, nset, reps)
        t = tline[label][0] * 1e-6 if label in tline else t_iso
        rows.append(dict(op=name, kernel=kernel, us=round(t * 1e6, 2), us_isolated=round(t_iso * 1e6, 2), launches_per_step=launches,
                         tflops=round(flops / t / 1e12, 1), gbs=round(nbytes / t / 1e9, 1),
                         mfma_frac=round(flops / t / 1e12 / peak_tflops, 4), hbm_frac=round(nbytes / t / 1e9 / peak_gbs, 4),
                         step_share_us=round(t * 1e6 * launches, 1)))
    dom = max(rows, key=lambda r: r["step_share_us"])
    total = sum(r["step_share_us"] for r in rows)
    mfma_bound = dom["mfma_frac"] >= dom["hbm_frac"]
    ach = dom["tflops"] if mfma_bound else dom["gbs"]
    peak = peak_tflops if mfma_bound else peak_gbs
    return {"bound": "mfma" if mfma_bound else "hbm", "kernel": dom["kernel"], "op": dom["op"], "achieved": ach,
            "peak": peak, "unit": "TFLOP/s" if mfma_bound else "GB/s", "frac": round(ach / peak, 4), "traffic": None,
            "avg_us": dom["us"], "launches_per_step": dom["launches_per_step"],
            "mfma_frac": dom["mfma_frac"], "hbm_frac": dom["hbm_frac"],
            # `bound` names the nearer of the two roofs (the contract's enum); a kernel below half of BOTH is bound by
            # neither: its time is the latency of one workgroup's dependent phases
            "regime": "latency" if max(dom["mfma_frac"], dom["hbm_frac"]) < 0.5 else ("mfma" if mfma_bound else "hbm"),
            "timing": ("in-step HIP events (sitk_timeline) of an EAGER re-run of the step: interval from the previous kernel's end "
                       "to this kernel's end = duration + the dependent-launch gap, an upper bound that reads ~4 % above "
                       "rocprofv3's kernel durations; the rows' sum can therefore exceed ms_per_step of the graph replay")
            if tline else "isolated replay",
            "buffer_sets": nset, "encoder_kernel_sum_us": round(total, 1), "kernels": rows}
