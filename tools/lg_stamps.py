"""Timeline of one workgroup of the fused LayerNorm + to_qkv forward kernel (build with -DSITK_LG_STAMPS)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import ops, runtime as rt  # noqa: E402

dev = "cuda:0"
R, D, N = 64 * 321, 192, 576
x = torch.randn(R, D, device=dev)
w = (torch.randn(N, D, device=dev) * 0.07).bfloat16()
lw, lb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
for _ in range(5):
    ops.ln_gemm_fwd(x, lw, lb, w, "bf16")
torch.cuda.synchronize()
buf = (C.c_ulonglong * 128)()
fn = rt.lib.sitk_lg_debug_stamps
fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
assert fn(buf) == 0
names = ["start", "x landed", "LN done", "barrier", "frags", "chunk0", "chunk4", "loop", "drained"]
for wv in range(8):
    print(f"wave {wv}: " + "  ".join(f"{names[i]}={buf[wv * 16 + i]}" for i in range(9)))
