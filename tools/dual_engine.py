"""Round 6: two HALF-batch training steps on two streams against one whole-batch step, whole engines (each replays its step from
one hipGraph).  For the wide configurations (dims 384 / 768: multi-round GEMM / attention / LayerNorm kernels that fill every CU)
the question is whether the memory-bound kernels of one half (LayerNorm, epilogues) fill the gaps of the other's GEMMs.  Two
engines with their own weights stand in for "the same engine on two halves" (timing only).

    python tools/dual_engine.py --model small --patches 1280 --batch 32 [--task mpp]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import engine  # noqa: E402
from sitk.models.mpp import masked_patch_pretraining  # noqa: E402
from sitk.models.sit import SiT  # noqa: E402

MODELS = {"tiny": dict(dim=192, depth=12, heads=3, mlp_dim=768, dim_head=64), "small": dict(dim=384, depth=12, heads=6, mlp_dim=1536, dim_head=64),
          "base": dict(dim=768, depth=12, heads=12, mlp_dim=3072, dim_head=64)}


def make(a, B, dev):
    V = {80: 561, 320: 153, 1280: 45}[a.patches]
    mk = MODELS[a.model]
    torch.manual_seed(1234)
    model = SiT(**mk, num_patches=a.patches, num_vertices=V, num_channels=4, compute_dtype="bf16")
    model.allow_synthetic_table = True
    if a.task == "mpp":
        model = masked_patch_pretraining(model, mk["dim"], 4 * V, "cpu", mask_prob=0.75, replace_prob=0.8, swap_prob=0.02, channels=4, num_vertices=V)
    eng = engine.TrainEngine(model, B, task=a.task, input_layout="surface", lr=1e-5, momentum=0.9, device=dev, use_graph=(None if a.default_form else True))
    g = torch.Generator(device=dev).manual_seed(100)
    x = torch.randn((B, 40962, 4), device=dev, generator=g)
    y = torch.randn((B,), device=dev, generator=g) * 2 + 40 if a.task == "regression" else None
    eng.load_batch(x, y)
    return eng


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="small")
    ap.add_argument("--patches", type=int, default=1280)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--task", default="regression")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--default-form", action="store_true", help="the engines' default launch form (dim 192: eager + side stream) instead of hipGraph replay")
    ap.add_argument("--parts", type=int, default=2, help="how many equal parts of the batch, each on a stream of its own")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    streams = [engine.pick_bucket_stream(dev)[0] for _ in range(a.parts)]
    s1 = streams[0]
    with torch.cuda.stream(s1):
        whole = make(a, a.batch, dev)
        for _ in range(3):
            whole.step()
    parts = []
    for st in streams:
        with torch.cuda.stream(st):
            e = make(a, a.batch // a.parts, dev)
            for _ in range(3):
                e.step()
            parts.append(e)
    ha = parts[0]
    torch.cuda.synchronize()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    def run_whole():
        with torch.cuda.stream(s1):
            whole.step()

    def run_serial():
        with torch.cuda.stream(s1):
            for _ in range(a.parts):
                ha.step()

    def run_dual():
        for st, e in zip(streams, parts):
            with torch.cuda.stream(st):
                e.step()

    for rep in range(3):
        print(f"{a.model} {a.patches} patches {a.task}: whole batch {a.batch}: {timed(run_whole):.3f} ms   {a.parts} parts on {a.parts} streams: {timed(run_dual):.3f} ms   "
              f"(one part {a.parts} times in a row: {timed(run_serial):.3f} ms)", flush=True)


if __name__ == "__main__":
    main()
