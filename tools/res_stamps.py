"""Phase timeline of the merged sequence-resident attention backward kernel (attn_bwd_res_kernel, N = 321), diagnostic build:

    make -C surface-vision-transformers_amd/csrc AB=1
    SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so python tools/res_stamps.py [--warm]

s_memtime stamps of four workgroups from the middle of the grid, every wave, in cycles since the workgroup's first stamp.
Columns: K / V / Wo DMA issued | landed + barrier | first query tile done | second query tile done (waves 0 - 4) |
workgroup barrier between the halves | Q / dO DMA issued, statistics stored | landed + barrier | first key tile done | second key
tile done (waves 0 - 4) | all stores drained.
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import ops  # noqa: E402
from sitk import runtime as rt  # noqa: E402

COLS = ["dma issued", "landed+bar", "q tile 1", "q tile 2", "mid barrier", "B issued", "B landed+bar", "k tile 1", "k tile 2", "end"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--heads", type=int, default=3)
    ap.add_argument("--warm", action="store_true", help="operands rewritten right before the launch (as inside the backward chain) instead of flushed")
    a = ap.parse_args()
    dev, dt = "cuda:0", "bf16"
    B, N, H, D = a.batch, 321, a.heads, 192
    I, R = H * 64, B * N
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(R, 3 * I, device=dev, generator=g).to(torch.bfloat16)
    dxmid = torch.randn(R, D, device=dev, generator=g).to(torch.bfloat16)
    wo_t = (torch.randn(I, D, device=dev, generator=g) * D ** -0.5).to(torch.bfloat16)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    o, lse = ops.attention_fwd(qkv, B, N, H, 0.125, dt)
    src = (qkv.clone(), dxmid.clone(), o.clone())
    for rep in range(3):
        flush.zero_()
        if a.warm:
            qkv.copy_(src[0]); dxmid.copy_(src[1]); o.copy_(src[2])
        torch.cuda.synchronize()
        ops.attention_bwd_proj(qkv, o, dxmid, wo_t, lse, B, N, H, 0.125, dt)
        torch.cuda.synchronize()
    lib = ctypes.CDLL(rt.LIB_PATH)
    lib.sitk_debug_res_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.sitk_debug_res_stamps.restype = ctypes.c_int
    st = np.zeros((4, 16, 12), dtype=np.uint64)
    rc = lib.sitk_debug_res_stamps(st.ctypes.data, st.nbytes)
    assert rc == 0, rc
    st = st.astype(np.int64)
    print(f"attn_bwd_res_kernel<FOLD> B {B} H {H} N {N}, operands {'just written' if a.warm else 'flushed'}; cycles since the workgroup's "
          "first stamp; columns: " + " | ".join(COLS))
    for w in range(4):
        t0 = st[w][:, 0][st[w][:, 0] > 0].min()
        print(f"workgroup {w}:")
        for wave in range(16):
            row = st[w, wave, 1:11]
            print(f"  wave {wave:2d}: " + " ".join(f"{int(x - t0):7d}" if x else "      -" for x in row))


if __name__ == "__main__":
    main()
