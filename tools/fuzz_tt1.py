"""One-off cross-check of the 12-wave (16-token) fused kernels against the 6-wave (32-token) ones they replaced: the two
geometries do the same arithmetic in the same order, so every output must be BIT-IDENTICAL (measured: all 196 row-indexed
tensors of 16 row counts from 1 to 24 576 are; the dgamma / dbeta partial sums differ in fp32 summation order only).  The switch is read once per
process: run `python tools/fuzz_tt1.py dump a.pt` under SITK_MLP_TT1=0 SITK_LG_TT1=0, `... dump b.pt` under =1, then
`python tools/fuzz_tt1.py cmp a.pt b.pt`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROWS = [1, 15, 16, 17, 95, 96, 97, 191, 193, 321, 642, 1000, 2049, 5184, 20544, 24576]


def dump(path):
    import sitk  # noqa: F401
    from sitk import ops
    dev, D, M, N3 = "cuda:0", 192, 768, 576
    out = {}
    g = torch.Generator(device=dev).manual_seed(7)
    rn = lambda *s, sc=0.5: torch.randn(*s, device=dev, generator=g) * sc  # noqa: E731
    w1, w2, wq = rn(M, D, sc=0.07).bfloat16(), rn(D, M, sc=0.04).bfloat16(), rn(N3, D, sc=0.07).bfloat16()
    b1, b2, lw, lb = rn(M, sc=0.1), rn(D, sc=0.1), rn(D, sc=0.3) + 1, rn(D, sc=0.2)
    for R in ROWS:
        x, dy = rn(R, D, sc=1.0), rn(R, D, sc=1.0)
        o, h, mean, rstd, u, gg = ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, "bf16", want_g=True)
        dx, dxc, du, part = ops.mlp_bwd(dy, dy.bfloat16(), x, mean, rstd, lw, w2.T.contiguous(), w1.T.contiguous(), u, "bf16")
        dqkv = rn(R, N3, sc=1.0).bfloat16()
        dx2, dxc2, part2 = ops.ln_gemm_bwd(dqkv, wq.T.contiguous(), x, mean, rstd, lw, dy, "bf16")
        torch.cuda.synchronize()
        for k, v in dict(o=o, h=h, mean=mean, rstd=rstd, u=u, g=gg, dx=dx, dxc=dxc, du=du, psum=part.sum(0), dx2=dx2,
                         dxc2=dxc2, psum2=part2.sum(0)).items():
            out[f"{R}/{k}"] = v.float().cpu()
    torch.save(out, path)
    print("dumped", len(out), "tensors to", path)


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    bad = 0
    for k in A:
        if k.endswith("psum") or k.endswith("psum2"):          # dgamma / dbeta partials: rows are summed per wave, 8 instead of 16 rows
            ok = bool(((A[k] - B[k]).abs() <= 1e-5 * A[k].abs().max()).all())   # per wave -> a different fp32 order (measured 1e-7 relative)
        else:
            ok = torch.equal(A[k], B[k])
        if not ok:
            bad += 1
            print("DIFF", k, float((A[k] - B[k]).abs().max()))
    print("compared", len(A), "tensors:", "all bit-identical" if bad == 0 else f"{bad} differ")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    dump(sys.argv[2]) if sys.argv[1] == "dump" else cmp(sys.argv[2], sys.argv[3])
