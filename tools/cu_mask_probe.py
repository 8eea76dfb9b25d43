"""Experiment: CU-masked streams (hipExtStreamCreateWithCUMask) -- can the weight gradients run on the CUs the
one-wave kernels of the main chain leave idle (214 of 256 workgroup slots used, 192 in attention)?

    python tools/cu_mask_probe.py
Mask bit i addresses XCD i % 8 (KFD spreads the mask round-robin over the XCCs), so the first 8 k bits give k CUs per XCD.
"""
import ctypes, glob, os, sys, time
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib
pkg = importlib.import_module("surface-vision-transformers_amd")
ops = importlib.import_module("surface-vision-transformers_amd.ops")


def hip():
    for p in glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so*")):
        return ctypes.CDLL(p)
    return ctypes.CDLL("libamdhip64.so")


def masked_stream(lib, lo, hi, nbits=256):
    words = (ctypes.c_uint32 * (nbits // 32))()
    for i in range(lo, hi):
        words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = lib.hipExtStreamCreateWithCUMask(ctypes.byref(s), nbits // 32, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def timed(fn, streams, reps=10):
    for s in streams:
        s.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    for s in streams:
        s.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    lib = hip()
    dev, dt, td = "cuda:0", "bf16", torch.bfloat16
    B, N, D, H = 64, 321, 192, 3
    M, I, R = 4 * D, H * 64, B * N
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, dtype=td: (torch.randn(*s, device=dev, generator=g) * 0.5).to(dtype)  # noqa: E731
    qkv = rn(R, 3 * I)
    o_att, lse = ops.attention_fwd(qkv, B, N, H, 0.125, dt)
    h, o, u, dxT = rn(R, D), rn(R, I), rn(R, M), rn(R, D)
    f32 = torch.float32
    dW = {k: torch.zeros(s, dtype=f32, device=dev) for k, s in dict(qkv=(3 * I, D), o=(D, I), w1=(M, D), w2=(D, M)).items()}
    bD, bM = torch.zeros(D, device=dev), torch.zeros(M, device=dev)
    probs = [dict(dY=dxT, X=u, dW=dW["w2"], db=bD), dict(dY=u, X=h, dW=dW["w1"], db=bM),
             dict(dY=dxT, X=o, dW=dW["o"], db=bD), dict(dY=qkv, X=h, dW=dW["qkv"])]
    WS = torch.empty(256 << 20, dtype=torch.uint8, device=dev)

    def chain(n=24):
        for _ in range(n):
            ops.attention_fwd(qkv, B, N, H, 0.125, dt)

    def wg(layers):
        ops.gemm_wgrad_group(probs * layers, dt, workspace=WS)

    cur = torch.cuda.current_stream()
    full = masked_stream(lib, 0, 256)
    main216 = masked_stream(lib, 0, 216)
    half = masked_stream(lib, 0, 128)
    side40 = masked_stream(lib, 216, 256)
    for name, st in (("default", cur), ("mask 256", full), ("mask 216", main216), ("mask 128", half), ("mask 40 (216..255)", side40)):
        with torch.cuda.stream(st):
            chain(4)
            t = timed(lambda: chain(24), [st])
        print(f"24 x attention fwd (192 workgroups) on {name}: {t / 24:.1f} us each")
    for name, st, layers in (("default", cur, 12), ("mask 216", main216, 12), ("mask 40", side40, 2), ("mask 40", side40, 12)):
        with torch.cuda.stream(st):
            wg(layers)
            t = timed(lambda: wg(layers), [st], reps=5)
        print(f"weight gradients of {layers} layers on {name}: {t:.0f} us")
    # concurrency: chain on 216 CUs, weight gradients on the other 40
    def both():
        with torch.cuda.stream(side40):
            wg(2)
        with torch.cuda.stream(main216):
            chain(24)
    both()
    t = timed(both, [side40, main216], reps=5)
    print(f"chain (24 x attention fwd) on mask 216 || weight gradients of 2 layers on mask 40: {t:.0f} us")
    def both_unmasked():
        with torch.cuda.stream(full):
            wg(2)
        with torch.cuda.stream(cur):
            chain(24)
    both_unmasked()
    t = timed(both_unmasked, [full, cur], reps=5)
    print(f"the same on two unmasked streams: {t:.0f} us")


if __name__ == "__main__":
    main()
