"""Robustness check of the two-stream step: many steps on a NEW batch each (resident data set, host indices), prefetched gather
against the in-line one, the head's gradient sums on the side stream against the main one -- losses and parameters must stay bit-equal; then a long run of the bench configuration.

    python tools/stress_two_stream.py [--steps 300]
"""
import argparse
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import engine  # noqa: E402
from sitk.models.sit import SiT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    base = SiT(dim=192, depth=12, heads=3, mlp_dim=768, num_patches=320, num_vertices=153, num_channels=4, compute_dtype="bf16")
    S, B = 96, 64
    g = torch.Generator(device=dev).manual_seed(1)
    xs = torch.randn((S, 40962, 4), device=dev, generator=g)
    ys = torch.randn((S, 1), device=dev, generator=g) + 40
    picks = [torch.randperm(S, generator=torch.Generator().manual_seed(i))[:B].numpy() for i in range(a.steps)]
    out = []
    for prefetch, deferred in ((True, True), (False, True), (True, False)):
        eng = engine.TrainEngine(copy.deepcopy(base), B, input_layout="surface", lr=1e-4, momentum=0.9, prefetch_gather=prefetch,
                                 head_deferred=deferred)
        eng.load_dataset(xs, ys)
        losses = []
        t0 = time.time()
        for idx in picks:
            losses.append(eng.step(indices=idx).clone())
        torch.cuda.synchronize()
        print(f"prefetch {prefetch}, head sums on the side stream {deferred}: {a.steps} steps in {time.time() - t0:.2f} s, last loss {float(losses[-1]):.6f}", flush=True)
        out.append((torch.cat(losses), eng.fp.flat.clone()))
    same_l = all(torch.equal(out[0][0], o[0]) for o in out[1:])
    same_p = all(torch.equal(out[0][1], o[1]) for o in out[1:])
    print("losses bit-equal:", same_l, " parameters bit-equal:", same_p)
    assert same_l and same_p
    eng = engine.TrainEngine(copy.deepcopy(base), B, input_layout="surface", lr=1e-5, momentum=0.9)
    eng.load_batch(xs[:B], ys[:B])
    for _ in range(10):
        eng.step()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 3000
    for _ in range(n):
        eng.step()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"{n} steps of the bench configuration: {dt / n * 1e3:.4f} ms per step, loss {float(eng.loss):.6f}")


if __name__ == "__main__":
    main()
