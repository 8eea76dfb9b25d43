"""Host time to ENQUEUE one data-parallel step when a rank has only a few CPUs (an 8-rank node inside a 16-CPU share gives each
rank 2): the process pins itself to `--cpus` CPUs, runs the step of bench.py's --dp-form (eager two-slice form) or the
graph-per-segment form, and reports (a) the host time of eng.step() -- the call returns when everything is enqueued -- and (b)
the step time.  If (a) approaches (b) the GPU waits for the host.

    python tools/dp_host_enqueue.py [--cpus 2 --graph]                                   one rank, RCCL (the real backend's enqueue path)
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dp_host_enqueue.py --backend gloo
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpus", type=int, default=2)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--graph", action="store_true", help="the graph-per-segment form (use_graph=True) instead of the eager one")
    ap.add_argument("--steps", type=int, default=60)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    avail = sorted(os.sched_getaffinity(0))
    mine = avail[(rank * a.cpus) % len(avail):][:a.cpus] or avail[:a.cpus]
    os.sched_setaffinity(0, set(mine))
    torch.set_num_threads(a.cpus)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("NCCL_MAX_NCHANNELS", "32")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev, pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=False))
    else:
        dist.init_process_group("gloo")
    import sitk  # noqa: F401
    from sitk import engine
    from sitk.models.sit import SiT
    torch.manual_seed(1234)
    B = 64
    model = SiT(dim=192, depth=12, heads=3, mlp_dim=768, dim_head=64, num_patches=320, num_vertices=153, num_channels=4, compute_dtype="bf16")
    eng = engine.TrainEngine(model, B, input_layout="surface", lr=1e-5, momentum=0.9, process_group=dist.group.WORLD, device=dev,
                             use_graph=True if a.graph else None)
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    eng.load_batch(torch.randn((B, 40962, 4), device=dev, generator=g), torch.randn((B,), device=dev, generator=g) * 2 + 40)
    for _ in range(6):
        eng.step()
    torch.cuda.synchronize()
    host, t0 = 0.0, time.perf_counter()
    for i in range(a.steps):
        h0 = time.perf_counter()
        eng.step()
        host += time.perf_counter() - h0
        if i % 10 == 9:
            torch.cuda.synchronize()               # the queue never runs more than ten steps ahead
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f"rank {rank}/{world} on CPUs {mine} ({a.backend}, {'graph per segment' if eng.use_graph else 'eager'}, dp_side {eng.dp_side}): "
          f"host enqueue {host / a.steps * 1e3:.3f} ms per step, step {total / a.steps * 1e3:.3f} ms", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
