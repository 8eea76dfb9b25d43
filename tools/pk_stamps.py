"""Phase timeline of the unit-packed attention kernels (diagnostic build: make -C surface-vision-transformers_amd/csrc AB=1, SITK_LIB=.../libsitk_ab.so):
s_memtime stamps of the first 8 workgroups, every wave, relative to the workgroup's first stamp.

    SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so python tools/pk_stamps.py [--kernel fwd|dq|dkv] [--batch 64 --heads 3]
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

NAMES = {
    "fwd": ["start", "dma issued", "K landed (own)", "barrier", "sweep 1", "phase 1 end", "barrier B1", "V landed (own)", "barrier B2", "phase 2 end", "stores"],
    "dq": ["start", "dma issued", "Wo landed", "fold done", "V issued", "K landed+bar", "phase 1 end", "V landed+bar", "phase 2a", "late landed+bar", "phase 2b", "end"],
    "dkv": ["start", "dma issued", "Q landed+bar", "phase 1 end", "dO landed+bar", "phase 2a", "late landed+bar", "phase 2b", "phase 3", "end"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="fwd", choices=list(NAMES))
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--heads", type=int, default=3)
    ap.add_argument("--wgs", type=int, default=4)
    ap.add_argument("--warm", action="store_true", help="the operands are written by a copy kernel right before the launch (as inside the forward chain) instead of flushed")
    a = ap.parse_args()
    dev, dt = "cuda:0", "bf16"
    B, N, H = a.batch, 321, a.heads
    I, R = H * 64, B * N
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(R, 3 * I, device=dev, generator=g).to(torch.bfloat16)
    d_o = torch.randn(R, I, device=dev, generator=g).to(torch.bfloat16)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    qkv_src, d_o_src = qkv.clone(), d_o.clone()
    for _ in range(3):
        o, lse = ops.attention_fwd(qkv, B, N, H, 0.125, dt)
    delta, dqkv = torch.zeros_like(lse), torch.empty_like(qkv)
    ki = {"fwd": 0, "dq": 1, "dkv": 2}[a.kernel]
    for rep in range(3):
        if a.warm:
            flush.zero_()
            qkv.copy_(qkv_src)
            d_o.copy_(d_o_src)
        else:
            flush.zero_()                  # cold caches, as inside the backward chain (the operands were written many kernels ago)
        torch.cuda.synchronize()
        if a.kernel == "fwd":
            ops.attention_fwd(qkv, B, N, H, 0.125, dt)
        else:
            rt.check(rt.lib.sitk_attention_bwd_phases(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), None, None, None, lse.data_ptr(),
                                                      delta.data_ptr(), dqkv.data_ptr(), B, N, H, I, 0.125, rt.BF16,
                                                      1 if a.kernel == "dq" else 2, rt.stream_ptr()))
        torch.cuda.synchronize()
    lib = ctypes.CDLL(rt.LIB_PATH)
    lib.sitk_debug_pk_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.sitk_debug_pk_stamps.restype = ctypes.c_int
    st = np.zeros((3, 8, 16, 12), dtype=np.uint64)
    rc = lib.sitk_debug_pk_stamps(st.ctypes.data, st.nbytes)
    assert rc == 0, rc
    st = st[ki].astype(np.int64)
    names = NAMES[a.kernel]
    print("kernel", a.kernel, "B", B, "H", H, "-- cycles since the workgroup's first stamp; columns:", ", ".join(names))
    for w in range(a.wgs):
        t0 = st[w][:, 0][st[w][:, 0] > 0].min()
        print(f"workgroup {w}:")
        for wave in range(16):
            row = st[w, wave, :len(names)] - t0
            rt_ticks = int(st[w, wave, 11])        # 100 MHz ticks of the wave's lifetime -> the shader clock it ran at
            life = int(row[len(names) - 1] - row[0])
            print(f"  wave {wave:2d}: " + " ".join(f"{int(x):7d}" for x in row) + (f"   | {rt_ticks * 10} ns, {life / (rt_ticks * 10.0):.2f} GHz" if rt_ticks else ""))


if __name__ == "__main__":
    main()
