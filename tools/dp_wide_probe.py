"""Round 6: what does the data-parallel FORM cost a wide model per rank (one-rank RCCL group, no wire)?  SiT-base MPP, 1280 patches,
B = 32 (BASELINE config 5's per-GPU share): the plain engine against the data-parallel engine with 1 / 2 / 3 backward slices and both
collective arrangements.

    python tools/dp_wide_probe.py [--model base --task mpp --batch 32 --steps 8]
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import engine  # noqa: E402
from sitk.models.mpp import masked_patch_pretraining  # noqa: E402
from sitk.models.sit import SiT  # noqa: E402

MODELS = {"tiny": dict(dim=192, depth=12, heads=3, mlp_dim=768, dim_head=64), "small": dict(dim=384, depth=12, heads=6, mlp_dim=1536, dim_head=64),
          "base": dict(dim=768, depth=12, heads=12, mlp_dim=3072, dim_head=64)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="base")
    ap.add_argument("--patches", type=int, default=1280)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--task", default="mpp")
    ap.add_argument("--steps", type=int, default=8)
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    V = {80: 561, 320: 153, 1280: 45}[a.patches]
    mk = MODELS[a.model]
    g = torch.Generator(device=dev).manual_seed(100)
    x = torch.randn((a.batch, 40962, 4), device=dev, generator=g)
    y = torch.randn((a.batch,), device=dev, generator=g) * 2 + 40 if a.task == "regression" else None
    for label, kw in (("plain engine", {}), ("process group, 1 slice", dict(process_group=dist.group.WORLD, bwd_slices=1)),
                      ("process group, 2 slices", dict(process_group=dist.group.WORLD, bwd_slices=2)),
                      ("process group, 3 slices (default)", dict(process_group=dist.group.WORLD)),
                      ("process group, 3 slices, eager", dict(process_group=dist.group.WORLD, use_graph=False)),
                      ("process group, 3 slices, async collectives", dict(process_group=dist.group.WORLD, dp_collective="group"))):
        torch.manual_seed(1234)
        model = SiT(**mk, num_patches=a.patches, num_vertices=V, num_channels=4, compute_dtype="bf16")
        model.allow_synthetic_table = True
        if a.task == "mpp":
            model = masked_patch_pretraining(model, mk["dim"], 4 * V, "cpu", mask_prob=0.75, replace_prob=0.8, swap_prob=0.02, channels=4, num_vertices=V)
        eng = engine.TrainEngine(model, a.batch, task=a.task, input_layout="surface", lr=1e-5, momentum=0.9, device=dev, **kw)
        eng.load_batch(x, y)
        for _ in range(3):
            eng.step()
        torch.cuda.synchronize()
        t0, host = time.perf_counter(), 0.0
        for _ in range(a.steps):
            h0 = time.perf_counter()
            eng.step()
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        print(f"{label:48s}: {ms:8.3f} ms per step (host enqueue {host / a.steps * 1e3:.2f} ms), slices {eng.slices}, graph {eng.use_graph}, "
              f"buckets {[sum(hi - lo for lo, hi in b) * 4 // 1000000 for b in eng.bucket_plan]} MB", flush=True)
        del eng, model
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
