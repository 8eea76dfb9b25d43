"""Per-phase cycle breakdown of the fused MLP forward kernel (diagnostic build, SITK_MLP_VAR=6)."""
import ctypes as C
import os
import sys

import torch

os.environ["SITK_MLP_VAR"] = "6"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import ops, runtime as rt  # noqa: E402

dev = "cuda:0"
R, D, M = 64 * 321, 192, 768
x = torch.randn(R, D, device=dev)
w1 = (torch.randn(M, D, device=dev) * 0.07).bfloat16()
w2 = (torch.randn(D, M, device=dev) * 0.04).bfloat16()
b1, b2, lw, lb = torch.zeros(M, device=dev), torch.zeros(D, device=dev), torch.ones(D, device=dev), torch.zeros(D, device=dev)
bwd = len(sys.argv) > 1 and sys.argv[1] == "bwd"
tail = len(sys.argv) > 1 and sys.argv[1] == "tail"      # the block-tail kernel: to_out + norm + MLP + next block's norm + to_qkv
out, h, mean, rstd, u, g = ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, "bf16")
dy = torch.randn(R, D, device=dev)
dyc = dy.bfloat16()
w2t, w1t = w2.T.contiguous(), w1.T.contiguous()
o_att = torch.randn(R, D, device=dev).bfloat16()
wo = (torch.randn(D, D, device=dev) * 0.07).bfloat16()
wq = (torch.randn(3 * D, D, device=dev) * 0.07).bfloat16()
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for _ in range(5):
    big.zero_()                                    # push the operands out of the caches, as in a real step
    if bwd:
        ops.mlp_bwd(dy, dyc, x, mean, rstd, lw, w2t, w1t, u, "bf16")
    elif tail:
        ops.attn_out_mlp_next_fwd(o_att, wo, b2, x, lw, lb, w1, b1, w2, b2, lw, lb, wq, "bf16", want_g=True)
    else:
        ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, "bf16")
torch.cuda.synchronize()
buf = (C.c_ulonglong * 320)()
NWAVES = 12 if os.environ.get("SITK_MLP_TT1", "0") != "0" else 8
fn = rt.lib.sitk_mlp_debug_stamps
fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
assert fn(buf) == 0
names = ["vmcnt wait", "barrier", "product 1", "elementwise", "product 2", "PROLOGUE", "EPILOGUE", "loop back"]
print("cycles (loop phases summed over the 12 chunks), workgroup 80 (s_memtime ticks):")
for w in range(NWAVES):
    print(f"wave {w}: " + "  ".join(f"{names[i]}={buf[w * 8 + i]}" for i in (5, 7, 0, 1, 2, 3, 4, 6)) +
          f"  total={sum(buf[w * 8 + i] for i in range(8))}")
if tail:
    print("block-tail timeline: [Wo + o landed, barrier] / [projection MFMAs done] / [residual + LayerNorm rows done] / [loop starts] (cycles since kernel start);"
          " after the loop: [pair exchange done] / [residual + next LayerNorm rows done] (cycles since the loop's end; EPILOGUE = all of it incl. the to_qkv loop)")
    for w in range(NWAVES):
        print(f"wave {w}: " + "  ".join(str(buf[128 + w * 8 + i]) for i in (5, 1, 2, 3)) + "   |   " + "  ".join(str(buf[128 + w * 8 + i]) for i in (6, 7)))
    print("appended to_qkv loop, summed over its 9 chunks: vmcnt wait / barrier / fragment reads + MFMAs / pack + store")
    for w in range(NWAVES):
        print(f"wave {w}: " + "  ".join(str(buf[224 + w * 4 + i]) for i in range(4)))
elif not bwd:
    print("forward prologue timeline (cycles since kernel start): before tables / before x loads / LN done / after barrier / fragments loaded")
    for w in range(NWAVES):
        print(f"wave {w}: " + "  ".join(str(buf[128 + w * 8 + i]) for i in range(5)))
