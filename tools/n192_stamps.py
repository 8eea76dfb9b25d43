"""Per-phase cycle breakdown of one workgroup of the N%192 GEMM kernel (diagnostic build: csrc/gemm.hip compiled with
-DSITK_N192_STAMPS into build/libsitk_stamps.so, selected by SITK_LIB).

    SITK_LIB=$GRAFT_REPO_ROOT/build/libsitk_stamps.so python tools/n192_stamps.py [--rows 40992 --n 1152 --k 384]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import ops, runtime as rt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=32 * 1281)
ap.add_argument("--n", type=int, default=1152)
ap.add_argument("--k", type=int, default=384)
a = ap.parse_args()
dev = "cuda:0"
x = (torch.randn(a.rows, a.k, device=dev) * 0.5).bfloat16()
w = (torch.randn(a.n, a.k, device=dev) * 0.05).bfloat16()
out = torch.empty(a.rows, a.n, device=dev, dtype=torch.bfloat16)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for _ in range(3):
    big.zero_()
    ops.gemm_nt(x, w, out, "bf16")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm_nt(x, w, out, "bf16"); e1.record(); torch.cuda.synchronize()
print(f"kernel {e0.elapsed_time(e1) * 1e3:.1f} us (instrumented build)")
buf = (C.c_ulonglong * 128)()
fn = rt.lib.sitk_n192_debug_stamps
fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
assert fn(buf) == 0
names = ["prologue", "dma wait", "barrier", "dma issue", "reads+mfma", "last barrier", "epilogue", "TOTAL"]
for wv in range(8):
    print(f"wave {wv}: " + "  ".join(f"{names[i]}={buf[wv * 8 + i]}" for i in range(8)))
