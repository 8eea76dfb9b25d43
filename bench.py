"""Headline benchmark: surfaces/s, forward + backward (+ SGD step), SiT-tiny, 320 patches, B = 64 per
GPU, bf16 MFMA, on N MI355X (BASELINE.json).  One "step" = one pass of the hot path over one batch
of synthetic surfaces already resident in HBM:

    gather (B,40962,4) -> patch embedding -> 12-layer encoder -> head -> MSE -> full backward
    -> [RCCL gradient all-reduce, N > 1] -> fused SGD(momentum 0.9) update

    python bench.py [--gpus N --steps K --warmup W]

N > 1: either already under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (one rank per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or typed bare -- then this process starts exactly that
launcher as a CHILD process (before anything touches the GPU), relays its output and exits with its return code.

Prints ONE JSON line (rank 0).  Extra objects:
  also         : (default N = 1 run only; skipped under a profiler and for A/B flags) the other numbers the docs quote, timed
                 behind the headline under the same clock -- the f16 mode IN this process with the headline's steps / warmup (the
                 parity-compliant figure), the headline with a new batch loaded every step, and as child processes (10 steps
                 each) the data-parallel form of the step and BASELINE configs 3 and 5
  roofline     : the dominant kernel (by time share in profiles/) timed live with HIP events on the launch
                 stream, with its algorithmic FLOPs per launch
  cpu_baseline : the CPU oracle (oracle/sit_oracle.py, "port") on this host's cores, config
                 BASELINE configs[0] (B = 4), bounded to ~15 s; at every N (rank 0, behind the timed region)
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODELS = {
    "tiny": dict(dim=192, depth=12, heads=3, mlp_dim=768, dim_head=64),
    "small": dict(dim=384, depth=12, heads=6, mlp_dim=1536, dim_head=64),
    "base": dict(dim=768, depth=12, heads=12, mlp_dim=3072, dim_head=64),
}
PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def step_gflop_per_sample(dim, depth, heads, mlp_dim, P, K, n_classes=1, mpp=False):
    """Algorithmic GEMM FLOPs fwd+bwd per sample: bwd = 2x fwd except the patch
    embedding, which needs no input gradient."""
    N, I = P + 1, heads * 64
    embed = 2 * P * K * dim
    layer = 2 * N * dim * 3 * I + 2 * 2 * heads * N * N * 64 + 2 * N * I * dim + 4 * N * dim * mlp_dim
    head = 2 * P * dim * K if mpp else 2 * dim * n_classes
    fwd = embed + depth * layer + head
    return (fwd + 2 * (fwd - embed) + embed) / 1e9


def host_cores():
    """CPU threads this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box shows 256
    logical CPUs but grants a 16-CPU share; 256 torch threads on it made the first bench run of round 1 time out inside
    the CPU baseline before it printed anything)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(seconds=15.0):
    """BASELINE configs[0]: SiT-tiny, 320 patches, B = 4, fp32, MSE, SGD(momentum 0.9) on the host.  Bounded: stops after
    `seconds` (3 timed steps at least) and in any case after 4 x `seconds`."""
    import numpy as np
    from oracle import sit_oracle
    cores = host_cores()
    torch.set_num_threads(cores)
    B = 4
    model = sit_oracle.SiT(**MODELS["tiny"], num_patches=320, num_vertices=153, num_channels=4)
    opt = torch.optim.SGD(model.parameters(), lr=1e-5, momentum=0.9)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((B, 4, 320, 153), generator=g)
    y = torch.randn((B,), generator=g)
    times = []
    t_start = time.perf_counter()
    t_end = t_start + seconds
    it = 0
    while True:
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(model(x).squeeze(), y)
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        if it >= 3:                       # three untimed warm-up steps
            times.append(dt)
        it += 1
        now = time.perf_counter()
        if (now > t_end and len(times) >= 3) or len(times) >= 200 or (now > t_start + 4 * seconds and times):
            break
    med = float(np.median(times))
    return {"value": B / med, "unit": "surfaces/s", "cores": cores, "kind": "port",
            "sample": f"oracle/sit_oracle.py SiT-tiny 320x153x4, B=4 fp32 fwd+bwd+SGD, median of {len(times)} steps "
                      f"({med * 1e3:.1f} ms/step), torch {torch.__version__} CPU"}


ALSO_SPECS = {
    # name: extra arguments of the child bench process (10 timed steps each, no probe, no CPU baseline)
    # (the f16 mode -- the 1e-3-compliant one -- is timed IN this process with the headline's own steps / warmup: time_engine)
    "dp_form": ["--dp-form"],                                    # what each rank of an N > 1 run executes (one-rank RCCL group)
    "cfg3": ["--model", "small", "--patches", "1280", "--batch", "32"],
    "cfg5": ["--model", "base", "--patches", "1280", "--batch", "32", "--task", "mpp"],
}


def also_lines(timeout_s=240):
    """The other configurations DESIGN.md / profiles/ quote, measured under the SAME driver clock as the headline: each one a
    child process of this bench (started after the headline's timed loop has ended; the parent idles meanwhile), its JSON line
    reduced to a few fields.  A child that fails or times out is reported as {"error": ...} -- never silently dropped."""
    out = {}
    # the children must not inherit a profiler's preload (rocprofv3 attaches through these): they would write their traces
    # into the parent's output directory
    env = {k: v for k, v in os.environ.items() if not (k == "HSA_TOOLS_LIB" or k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")))}
    if "LD_PRELOAD" in env:        # (only a profiler's own preload goes; whatever else the host preloads stays with the children)
        kept = [t for t in env["LD_PRELOAD"].replace(":", " ").split() if "rocprof" not in t and "roctracer" not in t]
        if kept:
            env["LD_PRELOAD"] = " ".join(kept) if " " in os.environ["LD_PRELOAD"] and ":" not in os.environ["LD_PRELOAD"] else ":".join(kept)
        else:
            del env["LD_PRELOAD"]
    for name, extra in ALSO_SPECS.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "3", "--no-probe", "--no-cpu-baseline",
               "--no-also"] + extra
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-200:]}"}
                continue
            d = json.loads(lines[-1])
            out[name] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "step_mfma_frac": d["step_mfma_frac"],
                         "dtype": d["dtype"], "steps": d["steps"], "workload": d["config"]["workload"],
                         "parallelism": d["config"]["parallelism"], "hip_graph": d["config"]["hip_graph"],
                         "batch_parts": d["config"].get("batch_parts", 1)}
        except subprocess.TimeoutExpired:
            out[name] = {"error": f"timeout after {timeout_s} s"}
    return out


def under_profiler():
    """A profiler preload in the environment (rocprofv3 / rocprof attach through these): the `also` object is skipped then --
    its extra steps and child processes would end up in the trace of the run that is being profiled."""
    return any(k in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES") or k.startswith(("ROCPROF_", "ROCPROFILER_")) for k in os.environ) or \
        "rocprof" in os.environ.get("LD_PRELOAD", "")


def self_launch_command(n, argv):
    """`python bench.py --gpus N` typed bare: the one-rank-per-GPU launcher to start as a child process."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="tiny", choices=list(MODELS))
    ap.add_argument("--patches", type=int, default=320, choices=[80, 320, 1280])
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--task", default="regression", choices=["regression", "mpp"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches")
    ap.add_argument("--graph", action="store_true", help="force hipGraph replay (no side stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0, help="time bound of the CPU oracle loop (rank 0)")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true", help="A/B: the patch gather on the main stream, in front of the patch embedding")
    ap.add_argument("--wgrad-overlap", type=int, default=None,
                    help="layers whose weight gradients run on a side stream beside the backward chain (default: engine's choice)")
    ap.add_argument("--no-head-deferred", action="store_true", help="A/B: the head's gradient sums behind the head kernel on the main stream")
    ap.add_argument("--overlap-cus", type=int, default=None, help="workgroups of one side-stream weight-gradient launch (default 42)")
    ap.add_argument("--dp-form", action="store_true",
                    help="N = 1 only: run the DATA-PARALLEL form of the step (dim 192: the one-GPU launch sequence + one all-reduce "
                         "bucket per side launch; other widths: 3 backward slices, one hipGraph per segment) with every bucket "
                         "all-reduced over a ONE-rank RCCL group -- what each rank of an N > 1 run executes, minus the wire time")
    ap.add_argument("--dp-channels", type=int, default=None,
                    help="RCCL channels (= workgroups of an all-reduce; NCCL_MAX_NCHANNELS) the step leaves CUs for (default 16)")
    ap.add_argument("--whole-batch", action="store_true", help="A/B: the plain engine where make_engine would split the batch over two streams")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the `also` object (f16, data-parallel form, configs 3 and 5 as child processes behind the headline)")
    ap.add_argument("--pg-priority", default="default", choices=["default", "high"],
                    help="priority of the RCCL process group's stream (see the comment where the group is created)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (gloo: rehearsal of the multi-rank path on a one-GPU box)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(subprocess.call(self_launch_command(args.gpus, sys.argv[1:])))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > 1 and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: local rank {local_rank} but {ndev} GPU(s) visible (RCCL needs one GPU per rank)")
    local_dev = local_rank % max(ndev, 1)              # gloo rehearsal: several ranks may share a GPU
    torch.cuda.set_device(local_dev)
    dev = torch.device(f"cuda:{local_dev}")
    pg = None
    if world > 1 or args.dp_form:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # Every kernel of the main chain is ONE wave of workgroups that each own a CU's LDS (214 of the 256 CUs, 192 in
        # attention): the all-reduce that overlaps backward has to fit in the CUs they leave idle, or each overlapped
        # kernel needs a second wave.  One RCCL channel = one workgroup; the 22 MB of gradients do not need more.
        # (round 5: 16; the buckets are 3.5 MB each now and the tail launch behind the chain leaves exactly that many CUs free)
        if args.dp_channels:
            os.environ["NCCL_MAX_NCHANNELS"] = str(args.dp_channels)
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
        args.dp_channels = int(os.environ["NCCL_MAX_NCHANNELS"])
        if args.backend == "nccl":
            # Priority of RCCL's stream.  Round 3 made it HIGH: the runtime keeps streams of different priorities on different
            # hardware queues, and a collective that waited for the side stream from the main stream's queue stalled the backward
            # chain (one-rank group: 3.14 ms per step against 2.65).  Round 4: the engine now issues every early bucket behind
            # an event from a stream of its own, after the whole chain has been enqueued, and a stand-in for a collective with
            # real wire time (tools/dp_cu_budget.py: 32 workgroups that hold their CUs for 161 + 88 us) cost +3.8 ms per step
            # on a HIGH-priority stream against +0.9 ms on a default one (profiles/r04_dp_budget.txt): pending high-priority
            # workgroups keep the chain's kernels from being dispatched.  Default priority is the default.
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=(args.pg_priority == "high"))
            dist.init_process_group("nccl", device_id=dev, pg_options=opts)
        else:
            dist.init_process_group("gloo")
        pg = dist.group.WORLD

    import sitk  # noqa: F401
    from sitk import engine
    from sitk.models.mpp import masked_patch_pretraining
    from sitk.models.sit import SiT

    V = {80: 561, 320: 153, 1280: 45}[args.patches]
    P, K, B = args.patches, 4 * V, args.batch
    mk = MODELS[args.model]
    torch.manual_seed(1234)                       # identical initial weights on every rank
    model = SiT(**mk, num_patches=P, num_vertices=V, num_channels=4, compute_dtype=args.dtype)
    model.allow_synthetic_table = True            # 1280 patches: synthetic table on synthetic surfaces (said in config.workload)
    if args.task == "mpp":
        model = masked_patch_pretraining(model, mk["dim"], K, "cpu", mask_prob=0.75, replace_prob=0.8, swap_prob=0.02,
                                         channels=4, num_vertices=V)
    # (engine.make_engine: the form measured fastest for the configuration -- SiT-small on one GPU runs its batch as two concurrent
    # half-batch steps on two streams, everything else the plain engine; --whole-batch forces the latter)
    mk_engine = engine.TrainEngine if args.whole_batch else engine.make_engine
    eng = mk_engine(model, B, task=args.task, input_layout="surface", lr=1e-5, momentum=0.9,
                    process_group=pg, use_graph=(True if args.graph else (False if args.no_graph else None)), device=dev,
                    wgrad_overlap=args.wgrad_overlap, prefetch_gather=not args.no_prefetch,
                    wgrad_overlap_cus=args.overlap_cus, head_deferred=not args.no_head_deferred,
                    dp_channels=args.dp_channels)
    g = torch.Generator(device=dev).manual_seed(100 + rank)   # every rank its own synthetic shard
    x = torch.randn((B, 40962, 4), device=dev, generator=g)
    y = torch.randn((B,), device=dev, generator=g) * 2 + 40 if args.task == "regression" else None
    eng.load_batch(x, y)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 2)):          # >= 2: first call runs eagerly, second captures nothing new
        eng.step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    loss = float(eng.loss)
    ms = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed
    gf = step_gflop_per_sample(mk["dim"], mk["depth"], mk["heads"], mk["mlp_dim"], P, K, mpp=args.task == "mpp")

    out = {
        "metric": "surfaces/sec fwd+bwd, SiT-tiny 320-patch, B=64, 1/2/4/8 MI355X",
        "value": round(value, 1), "unit": "surfaces/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"SiT-{args.model} {P} patches{' (synthetic table)' if P == 1280 else ''} x {V} vertices x 4 channels, B={B}/GPU, {args.task}: "
                               f"gather(B,40962,4) + fwd + {'masked MSE' if args.task == 'mpp' else 'MSE'} + bwd + "
                               f"SGD(m=0.9), " + {"bf16": "bf16 MFMA / fp32 accumulate", "f16": "f16 MFMA / fp32 accumulate, loss-scaled backward",
                                                  "f32": "f32 MFMA (verification mode)"}[args.dtype],
                   "global_batch": B * world, "parallelism": f"dp{world}" + (" (data-parallel form on a one-rank RCCL group)" if args.dp_form and world == 1 else ""),
                   "hip_graph": bool(eng.use_graph),
                   "batch_parts": len(getattr(eng, "parts", [eng])),      # 2: two concurrent half-batch steps (engine.SplitTrainEngine)
                   "wgrad_overlap_layers": int(eng.wgrad_overlap),
                   "loss_after": round(loss, 6),
                   # placement of the engine's extra streams, measured at construction (sitk_stream_probe): chain of dependent
                   # launches alone / with the stream blocked behind an event, us; ok = harmless and concurrent
                   "stream_probe": {k: [{kk: (round(vv, 1) if isinstance(vv, float) else vv) for kk, vv in r.items()} for r in v]
                                    for k, v in (("side", getattr(eng, "side_stream_probe", [])),
                                                 ("bucket", getattr(eng, "dp_stream_probe", [])),
                                                 ("parts", [r for pr in getattr(eng, "stream_probe", []) for r in pr])) if v}},
        "step_gflop_per_sample": round(gf, 3),
        "step_mfma_frac": round(value * gf / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
    }
    if rank == 0:
        if not args.no_probe:
            from sitk import probe
            # (a split engine: the kernels of ONE half-batch step, the shapes its launches really have)
            out["roofline"] = probe.dominant_kernel_roofline(getattr(eng, "owner", eng), PEAK_BF16_TFLOPS, PEAK_HBM_GBS)
            try:  # HBM traffic of the dominant kernel from the committed PMC run (cannot be collected live)
                tr = json.load(open(os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")))
                if args.model == "tiny" and args.batch == 64 and args.patches == 320:
                    for e in tr["entries"]:
                        if e["kernel"] == out["roofline"]["kernel"]:
                            out["roofline"]["traffic"] = e["hbm_bytes_per_launch"]
                            out["roofline"]["traffic_source"] = e["source"] + "; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes"
            except (OSError, KeyError, ValueError):
                pass
        headline = (world == 1 and args.model == "tiny" and args.patches == 320 and args.batch == 64 and args.task == "regression"
                    and args.dtype == "bf16" and not args.dp_form and not args.graph and not args.no_graph
                    and args.wgrad_overlap is None and args.overlap_cus is None and not args.no_prefetch and not args.no_head_deferred)
        if headline and not args.no_also and not under_profiler():
            # a new batch EVERY step (load_batch of device tensors + step): the prefetched gather then waits for the copy, i.e.
            # the form a training loop with a host-side loader runs; 20 steps behind 4 untimed ones, in this process
            x2 = torch.randn((B, 40962, 4), device=dev, generator=g)
            for i in range(4):                                   # (the first steps of the new form are not representative)
                eng.load_batch(x2 if i % 2 else x, y)
                eng.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(20):
                eng.load_batch(x2 if i % 2 else x, y)
                eng.step()
            torch.cuda.synchronize()
            nb_ms = (time.perf_counter() - t1) / 20 * 1e3
            out["also"] = {"new_batch_every_step": {"ms_per_step": round(nb_ms, 4), "value": round(B / nb_ms * 1e3, 1),
                                                    "step_mfma_frac": round(B / nb_ms * gf / PEAK_BF16_TFLOPS, 4)}}
            # The f16 compute mode -- the one that meets north_star's 1e-3 against the CPU oracle with FIXED bars (DESIGN.md
            # section 2; the bf16 headline's bars are 2 x its measured error) -- in THIS process, same batch, same steps / warmup,
            # same timing brackets as the headline: the parity-compliant throughput figure.
            del eng
            torch.manual_seed(1234)
            m16 = SiT(**mk, num_patches=P, num_vertices=V, num_channels=4, compute_dtype="f16")
            e16 = engine.TrainEngine(m16, B, task="regression", input_layout="surface", lr=1e-5, momentum=0.9, device=dev)
            e16.load_batch(x, y)
            for _ in range(max(args.warmup, 2)):
                e16.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                e16.step()
            torch.cuda.synchronize()
            f_ms = (time.perf_counter() - t1) / args.steps * 1e3
            out["also"]["f16"] = {"ms_per_step": round(f_ms, 4), "value": round(B / f_ms * 1e3, 1),
                                  "step_mfma_frac": round(B / f_ms * gf / PEAK_BF16_TFLOPS, 4), "dtype": "f16", "steps": args.steps,
                                  "warmup": args.warmup, "in_process": True, "loss_after": round(float(e16.loss), 6),
                                  "note": "parity-compliant figure: every f16 test bar is north_star's fixed 1e-3 (one documented "
                                          "exception, tests/parity_bars.py)"}
            del e16, m16
            out["also"].update(also_lines())
        if not args.no_cpu_baseline:
            # at EVERY N (north_star: "in the same run"): rank 0 times the oracle behind the timed region while the other ranks
            # sleep on the rendezvous store (a blocking socket read: no spinning thread competes for the host's cores)
            out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_seconds)
        print(json.dumps(out), flush=True)
    if pg is not None:
        from datetime import timedelta
        store = torch.distributed.distributed_c10d._get_default_store()
        if rank == 0:
            store.set("sitk_bench_rank0_done", "1")
        else:
            store.wait(["sitk_bench_rank0_done"], timedelta(seconds=900))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
