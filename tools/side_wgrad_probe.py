"""Can the weight gradients of finished layers run BESIDE the backward chain, on the 42 CUs its 214-workgroup kernels leave idle?
Main stream: 24 fused MLP backward launches (214 workgroups of 768 threads, one per CU).  Side stream(s): one layer's four
weight-gradient problems as 21 tiles over all tokens (sitk_gemm_wgrad_group_ws_cus, cus = 21: one workgroup per tile, no token
split), repeated.  Prints the main chain's time alone / beside 1 / beside 2 side streams, and the side launch's own time."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitk  # noqa: E402,F401
from sitk import ops, runtime as rt  # noqa: E402

dev = "cuda:0"
R, D, M, I = 64 * 321, 192, 768, 192
td = torch.bfloat16
x = torch.randn(R, D, device=dev)
w1 = (torch.randn(M, D, device=dev) * 0.07).to(td)
w2 = (torch.randn(D, M, device=dev) * 0.04).to(td)
b1, b2, lw, lb = torch.zeros(M, device=dev), torch.zeros(D, device=dev), torch.ones(D, device=dev), torch.zeros(D, device=dev)
out, h, mean, rstd, u, g = ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, "bf16", want_g=True)
dy = torch.randn(R, D, device=dev)
dyc = dy.to(td)
w2t, w1t = w2.T.contiguous(), w1.T.contiguous()


def layer_problems():
    du, dqkv, o, h1 = (torch.randn(R, n, device=dev).to(td) for n in (M, 3 * I, I, D))
    return [dict(dY=dyc, X=g, dW=torch.zeros(D, M, device=dev), db=torch.zeros(D, device=dev)),
            dict(dY=du, X=h, dW=torch.zeros(M, D, device=dev), db=torch.zeros(M, device=dev)),
            dict(dY=dyc, X=o, dW=torch.zeros(D, I, device=dev), db=torch.zeros(D, device=dev)),
            dict(dY=dqkv, X=h1, dW=torch.zeros(3 * I, D, device=dev))]


def desc_array(problems):
    arr = (rt.WgradDesc * len(problems))()
    for d, p in zip(arr, problems):
        dY, X, dW, db = p["dY"], p["X"], p["dW"], p.get("db")
        d.M, d.N, d.K = dY.shape[0], dW.shape[0], dW.shape[1]
        d.dY, d.lddy, d.dy_is_f32, d.dymap = dY.data_ptr(), dY.stride(0), 0, rt.RowMap(0, 0, 0)
        d.X, d.ldx, d.xmap = X.data_ptr(), X.stride(0), rt.RowMap(0, 0, 0)
        d.dW, d.lddw, d.db = dW.data_ptr(), dW.stride(0), rt.ptr(db)
    return arr


probs = [layer_problems() for _ in range(2)]
arrs = [desc_array(p) for p in probs]
nbytes = rt.lib.sitk_gemm_wgrad_group_ws_bytes(arrs[0], 4, rt.BF16)
wss = [torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev) for _ in range(2)]
side = [torch.cuda.Stream(), torch.cuda.Stream()]
main = torch.cuda.current_stream()


def side_launch(k, cus, reps):
    with torch.cuda.stream(side[k]):
        for _ in range(reps):
            rt.check(rt.lib.sitk_gemm_wgrad_group_ws_cus(arrs[k], 4, rt.BF16, wss[k].data_ptr(), wss[k].numel(), cus, rt.stream_ptr()))


def main_chain(n=24):
    for _ in range(n):
        ops.mlp_bwd(dy, dyc, x, mean, rstd, lw, w2t, w1t, u, "bf16")


def timed(fn):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3


main_chain(4)
for cus in (21, 16, 32):
    t_side = timed(lambda: (side_launch(0, cus, 1), main.wait_stream(side[0])))
    print(f"one layer's weight gradients alone, cus = {cus}: {t_side:.1f} us")
for rep in range(3):
    t0 = timed(main_chain)
    res = [f"alone {t0:.1f}"]
    for nside, cus in ((1, 21), (2, 21), (2, 16), (1, 42)):
        def both():
            for k in range(nside):
                side[k].wait_stream(main)
                side_launch(k, cus, 3)
            main_chain()
        t = timed(both)
        torch.cuda.synchronize()
        res.append(f"beside {nside} x cus {cus}: {t:.1f}")
    print("main chain of 24 MLP backward launches, us:  " + "  ".join(res))
