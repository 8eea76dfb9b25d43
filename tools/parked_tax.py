"""Round 6: what do streams that sit PARKED behind an event cost the headline step?  (tools/micro/blocked_queue.hip: even on a
harmless hardware queue one more parked stream costs a 110-kernel chain +80 us, ~0.7 us per dispatch; the step's own side stream
is parked behind fork events most of the time.)  The default one-GPU engine, K extra streams parked for the whole step: a helper
stream runs a one-workgroup 2-ms spin kernel and records an event, each extra stream waits for it.  Needs SITK_LIB=libsitk_ab.so.

    python tools/parked_tax.py [--steps 40]
"""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import engine  # noqa: E402
from sitk import runtime as rt  # noqa: E402
from sitk.models.sit import SiT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    lib = ctypes.CDLL(rt.LIB_PATH)
    lib.sitk_debug_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.sitk_debug_occupy.restype = ctypes.c_int
    torch.manual_seed(1234)
    model = SiT(dim=192, depth=12, heads=3, mlp_dim=768, dim_head=64, num_patches=320, num_vertices=153, num_channels=4, compute_dtype="bf16")
    eng = engine.TrainEngine(model, 64, input_layout="surface", lr=1e-5, momentum=0.9, device=dev)
    g = torch.Generator(device=dev).manual_seed(100)
    x = torch.randn((64, 40962, 4), device=dev, generator=g)
    y = torch.randn((64,), device=dev, generator=g) * 2 + 40
    eng.load_batch(x, y)
    helper = torch.cuda.Stream(device=dev, priority=0)
    pool = []
    for _ in range(6):
        st = torch.cuda.Stream(device=dev)
        r = engine.probe_stream(st)
        rs = engine.probe_stream(st, main=eng._side_torch)
        pool.append((st, r, rs))
        print(f"stream {st.cuda_stream:#x}: parked vs main {r['blocked_us'] / r['free_us']:.2f} x (its kernel done {r['done_us']:.0f} us), vs side "
              f"{rs['blocked_us'] / rs['free_us']:.2f} x")
    ev = torch.cuda.Event()
    main_s = torch.cuda.current_stream()

    def run(parked, spin_only=False):
        for _ in range(5):
            eng.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            if parked or spin_only:
                helper.wait_stream(main_s)                  # the spin starts with the step
                assert lib.sitk_debug_occupy(1, 2000, helper.cuda_stream) == 0
                ev.record(helper)
                for st in parked:
                    st.wait_event(ev)
            eng.step()
            if parked or spin_only:
                main_s.wait_event(ev)
                for st in parked:
                    main_s.wait_stream(st)
            if i % 10 == 9:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    base = run([])
    print(f"no extra stream: {base:.3f} ms per step")
    print(f"helper's 2-ms one-workgroup spin kernel alone (nothing parked): {run([], spin_only=True):.3f} ms per step (the step then lasts >= 2.0 ms + the optimizer)")
    for k in (1, 2, 3):
        for sel in range(0, len(pool) - k + 1, k):
            sts = [p[0] for p in pool[sel:sel + k]]
            ms = run(sts)
            print(f"{k} parked: streams {[hex(s.cuda_stream) for s in sts]}: {ms:.3f} ms per step ({(ms - base) * 1e3:+.0f} us)", flush=True)
    print(f"no extra stream again: {run([]):.3f} ms per step")


if __name__ == "__main__":
    main()
