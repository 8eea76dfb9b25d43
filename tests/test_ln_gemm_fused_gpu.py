"""GPU parity tests of the fused LayerNorm + to_qkv kernels (csrc/ln_gemm_fused.hip; PreNorm(LayerNorm,
Attention) up to the bias-free to_qkv, state-dict keys layers.i.0.{norm, fn.to_qkv}) against float64 torch
math with GEMM operands rounded to bf16 where the kernels round.  Tolerances: relative L2 error."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detgen  # noqa: E402

DEV = "cuda:0"


class _H:
    """The 16-bit compute type under test: every test of this module runs once per type (bf16, f16)."""
    name, td = "bf16", torch.bfloat16


@pytest.fixture(autouse=True, params=["bf16", "f16"])
def _h16(request):
    _H.name = request.param
    _H.td = torch.bfloat16 if request.param == "bf16" else torch.float16
    yield
    _H.name, _H.td = "bf16", torch.bfloat16

D = 192


@pytest.fixture(scope="module")
def ops():
    import sitk  # noqa: F401
    from sitk import ops as _ops
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _ops


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(name, shape, scale=1.0):
    return torch.from_numpy(detgen.normal(name, shape, std=scale, seed=0)).to(DEV)


def r16(t):
    return t.to(_H.td).double()


SHAPES = [(128, 64), (963, 576), (1000, 192), (20544, 576), (130, 1152), (25000, 64),   # 25 000 rows: 128-row workgroups (one round)
          (41088, 576)]                                                                  # B = 128: 96-row workgroups in TWO rounds


def test_supported(ops):
    assert ops.ln_gemm_fused_supported(192, 576, _H.name)
    assert not ops.ln_gemm_fused_supported(384, 1152, _H.name) and not ops.ln_gemm_fused_supported(192, 576, "f32")


@pytest.mark.parametrize("rows,N", SHAPES)
def test_ln_gemm_fwd(ops, rows, N):
    x = rnd("lg/x", (rows, D), 1.5)
    ln_w, ln_b = rnd("lg/lw", (D,), 0.3) + 1.0, rnd("lg/lb", (D,), 0.2)
    w = rnd("lg/w", (N, D), D ** -0.5)
    y, h, mean, rstd = ops.ln_gemm_fwd(x, ln_w, ln_b, w.to(_H.td), _H.name)
    torch.cuda.synchronize()
    xd = x.double()
    assert rel(h, torch.nn.functional.layer_norm(xd, (D,), ln_w.double(), ln_b.double(), 1e-5)) < 4e-3
    assert rel(mean, xd.mean(1)) < 1e-5
    assert rel(rstd, (xd.var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-5
    assert rel(y, h.double() @ r16(w).T) < 3e-3               # from the kernel's own bf16 h: one rounding of y
    y2 = ops.ln_gemm_fwd(x, ln_w, ln_b, w.to(_H.td), _H.name, save=False)[0]
    assert torch.equal(y, y2)


def test_ln_gemm_fwd_integer_exact(ops):
    """integer weights, LayerNorm output taken from the kernel: products exact in fp32 -> only y's rounding"""
    rows, N = 300, 576
    x = rnd("lgi/x", (rows, D), 1.0)
    w = torch.from_numpy(detgen.randint("lgi/w", (N, D), -2, 3).astype(np.float32)).to(DEV)
    y, h, _, _ = ops.ln_gemm_fwd(x, torch.ones(D, device=DEV), torch.zeros(D, device=DEV), w.to(_H.td), _H.name)
    ref = (h.double() @ w.double().T).to(_H.td)      # exact sum, then the same single rounding
    if _H.name == "bf16":
        assert torch.equal(y, ref.to(y.dtype))       # 8-bit h times small integers: every fp32 partial sum is exact
    else:                                            # 11-bit h: the fp32 accumulator rounds on the way, y may differ by one
        assert rel(y, ref) < 3e-4                    # f16 unit in the last place (2^-11 = 4.9e-4 per element at worst)


@pytest.mark.parametrize("rows,N", SHAPES)
@pytest.mark.parametrize("with_res", [True, False])
def test_ln_gemm_bwd(ops, rows, N, with_res):
    x = rnd("lgb/x", (rows, D), 1.5)
    ln_w, ln_b = rnd("lgb/lw", (D,), 0.3) + 1.0, rnd("lgb/lb", (D,), 0.2)
    w = rnd("lgb/w", (N, D), D ** -0.5)
    dy = rnd("lgb/dy", (rows, N), 1.0).to(_H.td)
    dres = rnd("lgb/dr", (rows, D), 1.0) if with_res else None
    _, h, mean, rstd = ops.ln_gemm_fwd(x, ln_w, ln_b, w.to(_H.td), _H.name)
    wt = w.to(_H.td).T.contiguous()                           # (D, N)
    dx, dx_c, partials = ops.ln_gemm_bwd(dy, wt, x, mean, rstd, ln_w, dres, _H.name)
    torch.cuda.synchronize()
    xd = x.double().requires_grad_(True)
    lw = ln_w.double().requires_grad_(True)
    lb = ln_b.double().requires_grad_(True)
    hd = torch.nn.functional.layer_norm(xd, (D,), lw, lb, 1e-5)
    hd.backward(dy.double() @ r16(w))
    ref = xd.grad + (dres.double() if with_res else 0)
    assert rel(dx, ref) < 2e-5
    assert rel(dx_c, ref) < 3e-3
    # rows per workgroup (fused_block_rows in csrc/fused_epilogue.h: fewer rounds of 256 workgroups x rows wins, ties to 96)
    blk = 96 if ((rows + 95) // 96 + 255) // 256 * 96 <= ((rows + 127) // 128 + 255) // 256 * 128 else 128
    assert partials.shape == ((rows + blk - 1) // blk, 2, D)
    assert rel(partials[:, 0].sum(0), lw.grad) < 2e-5
    assert rel(partials[:, 1].sum(0), lb.grad) < 2e-5


def test_repeatable_and_in_bounds(ops):
    from sitk import runtime as rt
    rows, N, pad = 20544 - 37, 576, 128
    x = rnd("lgr/x", (rows, D), 1.5)
    ln_w, ln_b = rnd("lgr/lw", (D,), 0.3) + 1.0, rnd("lgr/lb", (D,), 0.2)
    w = rnd("lgr/w", (N, D), D ** -0.5).to(_H.td)
    wt = w.T.contiguous()
    dy = rnd("lgr/dy", (rows, N), 1.0).to(_H.td)
    dres = rnd("lgr/dr", (rows, D), 1.0)
    ref = None
    for _ in range(4):
        y, h, mean, rstd = ops.ln_gemm_fwd(x, ln_w, ln_b, w, _H.name)
        got = (y, h) + ops.ln_gemm_bwd(dy, wt, x, mean, rstd, ln_w, dres, _H.name)
        torch.cuda.synchronize()
        if ref is None:
            ref = [t.clone() for t in got]
        else:
            for a, b in zip(ref, got):
                assert torch.equal(a, b)
    yb = torch.full((rows + pad, N), 7.0, dtype=_H.td, device=DEV)
    hb = torch.full((rows + pad, D), 7.0, dtype=_H.td, device=DEV)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    rt.check(rt.lib.sitk_ln_gemm_fwd(x.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), w.data_ptr(), hb.data_ptr(), mean.data_ptr(),
                                     rstd.data_ptr(), yb.data_ptr(), rows, D, N, rt.dtype_code(_H.name), rt.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(yb[:rows], ref[0]) and torch.equal(hb[:rows], ref[1])
    assert bool((yb[rows:].float() == 7.0).all()) and bool((hb[rows:].float() == 7.0).all())
