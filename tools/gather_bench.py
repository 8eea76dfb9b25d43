"""The patch gather by itself on DISTINCT batches: `--sets` input batches of (B, 40962, 4) fp32 (8 x 42 MB = 336 MB at B = 64:
more than the 256 MB Infinity Cache holds) gathered round-robin, so that every launch reads its surfaces from HBM as a
training loop with a new batch per step does (bench.py re-gathers ONE resident batch).  Prints us per launch and the
algorithmic GB/s (B x 40962 x 16 B read + B x P x ld x 2 B written).

    python tools/gather_bench.py [--batch 64 --sets 8 --reps 40]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import ops, tables  # noqa: E402
from sitk import runtime as rt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--sets", type=int, default=8)
    ap.add_argument("--reps", type=int, default=40)
    a = ap.parse_args()
    dev = "cuda:0"
    B, P, V = a.batch, 320, 153
    t = tables.table_tensor(tables.load_table(P, V), dev)
    ld = ops.pad64(V * 4)
    g = torch.Generator(device=dev).manual_seed(0)
    xs = [torch.randn((B, 40962, 4), device=dev, generator=g) for _ in range(a.sets)]
    out = torch.empty((B * P, ld), dtype=torch.bfloat16, device=dev)

    def launch(i):
        rt.check(rt.lib.sitk_gather_tokens(xs[i % a.sets].data_ptr(), t.data_ptr(), out.data_ptr(), B, 40962, 4, P, V, ld, rt.BF16,
                                           rt.stream_ptr()))
    for i in range(a.sets):
        launch(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.reps):
        launch(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.reps * 1e3
    nbytes = B * 40962 * 16 + B * P * ld * 2
    print(f"gather B={B} over {a.sets} distinct batches: {us:.1f} us per launch (incl. ~2 us launch gap), "
          f"{nbytes / us / 1e3:.0f} GB/s algorithmic = {nbytes / us / 1e3 / 8000 * 100:.1f} % of 8 TB/s "
          f"({nbytes / 1e6:.1f} MB: {B * 40962 * 16 / 1e6:.1f} read + {B * P * ld * 2 / 1e6:.1f} written)")


if __name__ == "__main__":
    main()
