"""Does the gradient all-reduce FIT beside the backward chain?  (round-3 review: with a one-rank process group the collective is
a copy -- the 2.46 ms "per-rank" step says nothing about the CUs and the time a real all-reduce takes.)

The data-parallel form of the step on a ONE-rank RCCL group (every collective on the real backend), plus, behind every
bucket's all-reduce, ON THE STREAM THE ENGINE'S COLLECTIVE RUNS ON, a DUMMY kernel of `--channels` workgroups x 256
threads that holds its CUs for the time the bucket needs on the wire: --latency-us + bytes / (--gbs GB/s) (SURVEY section 5:
22 MB in ~0.25 ms per ring = 88 GB/s; the mesh algorithm is ~7x faster; 30 us for the launch + the ring's hops of a small
message).  Round 5: the stand-in has the footprint of RCCL's own kernel on gfx950 (256 threads, 19 744 B of LDS, 280 registers:
csrc/core.hip), the step is the per-side-launch bucket form, and `--side-cus` / `--channels` take lists to sweep.

Round 6: the stand-in follows the engine's arrangement.  `--collective stream` (the engine's default since round 6: synchronous c10d
collectives issued with the bucket stream current, which torch >= 2.8 launches on THAT stream; the final bucket on the main stream)
puts the stand-in on the stream that is current when the engine issues the collective -- same number and kind of streams as an N > 1
run.  `--collective group` is the control: the engine issues async_op=True collectives (rounds 2 - 5), whose kernel runs on the
process group's internal stream behind an event, and the stand-in runs on a stream of its own behind an event in the same way
(`--own-stream` overrides the placement alone).  `--skip-streams n` takes n streams from torch's pool first, so that the stand-in's
own stream lands on another hardware queue (ROCm maps streams onto GPU_MAX_HW_QUEUES queues in creation order).  Needs the
diagnostic build:

    SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so python tools/dp_cu_budget.py [--channels 16 --side-cus 42,34,26]
"""
import argparse
import ctypes
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import engine  # noqa: E402
from sitk import runtime as rt  # noqa: E402
from sitk.models.sit import SiT  # noqa: E402


class _EventWait:
    """Stands in the engine's list of pending work handles: wait() puts the current stream behind the stand-in's end."""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", default="16", help="comma list: workgroups of the stand-in = NCCL_MAX_NCHANNELS the engine plans for")
    ap.add_argument("--side-cus", default="42", help="comma list: workgroups of one side launch of weight gradients")
    ap.add_argument("--configs", default=None, help="instead of the two lists: layers:per_launch:workgroups:channels,... (layers on the side "
                    "stream, layers per side launch, workgroups a side launch is planned for, stand-in channels)")
    ap.add_argument("--latency-us", type=float, default=30.0, help="fixed part of a bucket's stand-in time")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--main-prio", type=int, default=0, help="-1: run the step itself on a HIGH-priority stream (main high, all-reduce path "
                    "normal, side stream lowest: three priority classes = three sets of hardware queues)")
    ap.add_argument("--collective", default="stream", choices=("stream", "group"), help="TrainEngine(dp_collective=...): 'stream' = the "
                    "shipped arrangement; 'group' = async_op=True on the process group's stream, the control")
    ap.add_argument("--own-stream", type=int, default=None, help="1: the stand-in runs on a stream of its own behind an event (where an "
                    "async_op=True collective runs); 0: on the stream that is current when the engine issues the collective (where a "
                    "synchronous one runs).  Default: follows --collective")
    ap.add_argument("--final-on", default="bucket", choices=("bucket", "main"), help="TrainEngine(dp_final_on=...): the final bucket from the bucket stream (default) or the main stream")
    ap.add_argument("--skip-streams", type=int, default=0, help="take this many streams from torch's pool before the stand-in's own one")
    ap.add_argument("--per-bucket", default=None, help="side launches per early all-reduce bucket: an int, or a comma list of bucket sizes "
                    "(launches it does not cover travel with the final bucket); engine default: all side launches in one early bucket")
    ap.add_argument("--standin-us", type=float, default=None, help="fixed stand-in time per bucket instead of latency + bytes / rate")
    ap.add_argument("--timeline", action="store_true", help="HIP-event timeline of three steps (no profiler attached): when each bucket is "
                    "ready, when its stand-in starts / ends, when the chain, the finish stage and the optimizer end")
    ap.add_argument("--gbs", type=float, default=88.0, help="wire rate of one bucket's all-reduce, GB/s of gradient bytes")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--prio", type=int, default=0, help="priority of the stream the dummy runs on (0 = default, like the process group of bench.py; -1 = high)")
    a = ap.parse_args()
    if a.own_stream is None:
        a.own_stream = int(a.collective == "group")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    chans = [int(c) for c in a.channels.split(",")]
    sides = [int(c) for c in a.side_cus.split(",")]
    os.environ.setdefault("NCCL_MAX_NCHANNELS", str(max(chans)))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev, pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=(a.prio < 0)))
    lib = ctypes.CDLL(rt.LIB_PATH)
    lib.sitk_debug_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.sitk_debug_occupy.restype = ctypes.c_int
    _skipped = [torch.cuda.Stream(device=dev, priority=a.prio) for _ in range(a.skip_streams)]
    hp = torch.cuda.Stream(device=dev, priority=a.prio)
    B = 64
    print(f"# collectives: {a.collective}; stand-in on {'a stream of its own behind an event' if a.own_stream else 'the stream the collective is issued on'}"
          f"; latency {a.latency_us:.0f} us + bytes / {a.gbs:.0f} GB/s per bucket, stream priority {a.prio}, {a.steps} steps, {a.dtype}, "
          f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(default)')}, {a.skip_streams} pool streams skipped")
    if a.configs:
        cfgs = [tuple(int(v) for v in c.split(":")) for c in a.configs.split(",")]
        cfgs = [c2 for c in cfgs for c2 in ((c[0], c[1], c[2], 0), c)]
    else:
        cfgs = [(None, 2, side, ch) for side in sides for ch in [0] + chans]
    for nlay, group, side, ch in cfgs:
      if True:
        torch.manual_seed(1234)
        model = SiT(dim=192, depth=12, heads=3, mlp_dim=768, dim_head=64, num_patches=320, num_vertices=153, num_channels=4,
                    compute_dtype=a.dtype)
        eng = engine.TrainEngine(model, B, input_layout="surface", lr=1e-5, momentum=0.9, process_group=dist.group.WORLD, device=dev,
                                 wgrad_overlap_cus=side, dp_channels=(ch or chans[0]), wgrad_overlap=nlay, wgrad_overlap_group=group,
                                 dp_stream_priority=a.prio, dp_collective=a.collective, dp_final_on=a.final_on, dp_bucket_launches=(None if a.per_bucket is None else (int(a.per_bucket) if a.per_bucket.isdigit() else [int(v) for v in a.per_bucket.split(",")])))
        assert eng.dp_side, "expected the side-stream form"
        fmt = lambda rs: ", ".join(f"{r['blocked_us']:.0f}/{r['free_us']:.0f} us done {r['done_us']:.0f}" + (f" side {r['victim0_blocked_us']:.0f}/{r['victim0_free_us']:.0f}" if 'victim0_free_us' in r else "") + (' *' if r['chosen'] else '') for r in rs)
        print(f"  placement probes (chain blocked / free, candidate's kernel done; * = chosen): side stream [{fmt(eng.side_stream_probe)}]  "
              f"bucket stream [{fmt(eng.dp_stream_probe)}]", flush=True)
        log, tl = [], []
        if not ch and a.timeline:
            orig0 = eng._allreduce

            def ready_only(lo, hi, orig0=orig0):
                orig0(lo, hi)
                e = torch.cuda.Event(enable_timing=True)
                e.record(torch.cuda.current_stream())
                tl.append(("ready", e))
            eng._allreduce = ready_only
        if ch:
            orig = eng._allreduce

            def wrapped(lo, hi, orig=orig, eng=eng, log=log, ch=ch):
                orig(lo, hi)
                us = max(1, int(a.latency_us + (hi - lo) * 4 / (a.gbs * 1e3))) if a.standin_us is None else max(1, int(a.standin_us))
                ev0, ev1 = torch.cuda.Event(enable_timing=a.timeline), torch.cuda.Event(enable_timing=a.timeline)
                cur = torch.cuda.current_stream()
                ds = hp if a.own_stream else cur
                if a.own_stream or a.timeline:
                    ev0.record(cur)
                if a.own_stream:
                    hp.wait_event(ev0)
                if a.timeline:
                    evs = torch.cuda.Event(enable_timing=True)
                    evs.record(ds)
                    tl.append(("ready", ev0)); tl.append(("stand-in start", evs)); tl.append(("stand-in end", ev1))
                if a.standin_us is None or a.standin_us > 0:          # (--standin-us 0: the events alone, no kernel)
                    assert lib.sitk_debug_occupy(ch, us, ds.cuda_stream) == 0
                ev1.record(ds)
                if a.own_stream:                      # (on the issuing stream the engine's own join covers the stand-in)
                    eng._pending.append(_EventWait(ev1))
                log.append(((hi - lo) * 4, us))
            eng._allreduce = wrapped
        g = torch.Generator(device=dev).manual_seed(100)
        x = torch.randn((B, 40962, 4), device=dev, generator=g)
        y = torch.randn((B,), device=dev, generator=g) * 2 + 40
        torch.cuda.synchronize()
        if a.main_prio:
            torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=a.main_prio))
        eng.load_batch(x, y)
        for _ in range(5):
            eng.step()
        torch.cuda.synchronize()
        t0, host = time.perf_counter(), 0.0
        for i in range(a.steps):
            h0 = time.perf_counter()
            eng.step()
            host += time.perf_counter() - h0
            if i % 10 == 9:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        if not ch:
            base = ms
        if a.timeline:
            # three more steps with events on the main stream: step start, end of the backward call (chain + tail launch enqueued
            # behind it), end of the finish stage, end of the optimizer
            orig_fin, orig_opt, orig_seg = eng._finish_backward, eng._optimizer, eng._segment_fns

            def mark(label):
                e = torch.cuda.Event(enable_timing=True)
                e.record(torch.cuda.current_stream())
                tl.append((label, e))

            def seg_fns():
                fns = orig_seg()
                return [lambda fn=fn: (fn(), mark("backward call done (main stream)")) for fn in fns]
            eng._segment_fns = seg_fns
            eng._finish_backward = lambda: (orig_fin(), mark("finish stage done"))
            eng._optimizer = lambda: (orig_opt(), mark("optimizer done"))
            import ctypes as C
            cap = 4096
            ktl = rt.lib.sitk_timeline_create(cap) if os.environ.get("SITK_TIMELINE_SIDE") == "1" else None
            for st in range(3):
                tl.clear()
                torch.cuda.synchronize()
                if ktl:
                    rt.lib.sitk_timeline_reset(ktl)
                    eng.cfg.timeline = ktl
                mark("step start")
                eng.step()
                torch.cuda.synchronize()
                t0 = tl[0][1]
                print(f"  timeline of step {st} (us from the step's start; events in issue order):")
                for label, e in tl[1:]:
                    print(f"    {t0.elapsed_time(e) * 1e3:9.1f}  {label}")
                if ktl and st == 2:
                    us, lab = (C.c_float * cap)(), (C.c_char_p * cap)()
                    n = rt.lib.sitk_timeline_read(ktl, us, lab, cap)
                    acc, seg = 0.0, []
                    for i in range(n):
                        k = lab[i].decode()
                        if k == "begin":
                            seg.append("   | " + f"(gap {us[i]:.0f})")
                            continue
                        seg.append(f"{k} {us[i]:.1f}")
                    print("  main-stream kernels of step 2 (sitk_timeline: interval from the previous mark, us): " + ", ".join(seg))
            if ktl:
                eng.cfg.timeline = None
                rt.lib.sitk_timeline_destroy(ktl)
        nb = sum(len(b) for b in eng.bucket_plan)
        extra = "  buckets (bytes, stand-in us): " + ", ".join(f"({b}, {u})" for b, u in log[-nb:]) if ch else ""
        print(f"{eng.wgrad_overlap} side layers, {group} per launch, {side:3d} workgroups, stand-in {ch:2d} channels: {ms:.3f} ms per step ({(ms - base) * 1e3:+5.0f} us; host enqueue "
              f"{host / a.steps * 1e3:.3f} ms){extra}", flush=True)
        del eng, model
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
