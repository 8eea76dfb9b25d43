"""GPU parity tests of the fused LayerNorm + MLP (+ residual) kernels (csrc/mlp_fused.hip; the
PreNorm(LayerNorm, FeedForward) half of a vit_pytorch block, state-dict keys layers.i.1.*) against a
float64 torch statement whose GEMM operands are rounded to bf16 exactly where the kernels round.

Tolerances are relative L2 errors; one bf16 rounding is 2^-9 = 2e-3 per stored value.
"""
import math

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


def rnd(name, shape, scale=1.0, seed=0):
    return torch.from_numpy(detgen.normal(name, shape, std=scale, seed=seed)).to(DEV)


def ints(name, shape, lo=-3, hi=4):
    return torch.from_numpy(detgen.randint(name, shape, lo, hi).astype(np.float32)).to(DEV)


def r16(t):
    return t.to(_H.td).double()


def gelu_grad(u):
    """d/du [u Phi(u)] = Phi(u) + u phi(u) (float64); the fused forward saves this MINUS 1/2 instead of u (ABI 9)."""
    u = u.double()
    return 0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)


def params(tag, M):
    ln_w, ln_b = rnd(tag + "/lw", (D,), 0.3) + 1.0, rnd(tag + "/lb", (D,), 0.2)
    w1, b1 = rnd(tag + "/w1", (M, D), D ** -0.5), rnd(tag + "/b1", (M,), 0.1)
    w2, b2 = rnd(tag + "/w2", (D, M), M ** -0.5), rnd(tag + "/b2", (D,), 0.1)
    return ln_w, ln_b, w1, b1, w2, b2


SHAPES = [(128, 64), (321 * 3, 768), (1000, 256), (20544, 768), (25000, 128),    # 25 000 rows: 128-row workgroups (one round)
          (41088, 768)]                                                           # B = 128: 96-row workgroups in TWO rounds


def test_supported_shapes(ops):
    assert ops.mlp_fused_supported(192, 768, _H.name)
    assert not ops.mlp_fused_supported(384, 1536, _H.name)      # small / base use the unfused kernels
    assert not ops.mlp_fused_supported(192, 768, "f32")        # the exact-f32 verification mode too
    assert not ops.mlp_fused_supported(192, 2048, _H.name)
    from sitk import runtime as rt
    x = torch.zeros((8, 384), device=DEV)
    with pytest.raises(rt.SitkError):
        ops.mlp_fwd(x, x[0], x[0], torch.zeros((64, 384), device=DEV).to(_H.td), x[0, :64],
                    torch.zeros((384, 64), device=DEV).to(_H.td), x[0], _H.name)


@pytest.mark.parametrize("rows,M", SHAPES)
def test_mlp_fused_fwd(ops, rows, M):
    """rows not a multiple of the 128-row workgroup; one / several / twelve hidden chunks; the BASELINE
    config-2 shape (20544 x 768)."""
    x = rnd("mlpf/x", (rows, D), 1.5)
    ln_w, ln_b, w1, b1, w2, b2 = params("mlpf", M)
    out, h, mean, rstd, gd, g = ops.mlp_fwd(x, ln_w, ln_b, w1.to(_H.td), b1, w2.to(_H.td), b2, _H.name, want_g=True)
    torch.cuda.synchronize()
    xd = x.double()
    h_r = torch.nn.functional.layer_norm(xd, (D,), ln_w.double(), ln_b.double(), 1e-5)
    assert rel(h, h_r) < 4e-3
    assert rel(mean, xd.mean(1)) < 1e-5
    assert rel(rstd, (xd.var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-5
    u_r = h.double() @ r16(w1).T + b1.double()                 # from the kernel's own (bf16) h: isolates the product
    assert rel(gd, gelu_grad(u_r) - 0.5) < 4e-3                # the saved (centred) derivative: fp32 gelu'(u) - 1/2, one rounding
    assert float((gd.double() + 0.5 - gelu_grad(u_r)).abs().max()) < (6e-3 if _H.name == "bf16" else 1e-3)
    g_r = torch.nn.functional.gelu(u_r)
    assert rel(g, g_r) < 6e-3
    br_r = r16(g_r) @ r16(w2).T + b2.double()
    assert rel(out - x, br_r) < 6e-3                           # the branch alone (the residual would mask errors)
    assert rel(out, xd + br_r) < 2e-3
    # inference form (nothing saved) gives the same output bits
    out2 = ops.mlp_fwd(x, ln_w, ln_b, w1.to(_H.td), b1, w2.to(_H.td), b2, _H.name, save=False)[0]
    assert torch.equal(out, out2)


def test_mlp_fused_fwd_integer_exact(ops):
    """Small-integer weights: given the kernel's own bf16 h and g, both products are exact in fp32 --
    any fragment-layout or hidden-permutation mistake shows as an O(1) error."""
    M, rows = 128, 200
    x = rnd("mlpi/x", (rows, D), 1.0)
    ln_w, ln_b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    w1, w2 = ints("mlpi/w1", (M, D), -2, 3), ints("mlpi/w2", (D, M), -2, 3)
    b1, b2 = ints("mlpi/b1", (M,)), ints("mlpi/b2", (D,))
    out, h, mean, rstd, gd, g = ops.mlp_fwd(x, ln_w, ln_b, w1.to(_H.td), b1, w2.to(_H.td), b2, _H.name, want_g=True)
    torch.cuda.synchronize()
    u_r = h.double() @ w1.double().T + b1.double()            # exact in fp32: integer weights, 16-bit h
    assert rel(gd, gelu_grad(u_r) - 0.5) < 3e-3                # one 16-bit rounding of the stored (centred) derivative
    assert rel(g, torch.nn.functional.gelu(u_r)) < 3e-3
    out_r = x.double() + g.double() @ w2.double().T + b2.double()
    assert rel(out, out_r) < 3e-6                              # fp32 accumulation of exactly representable products


@pytest.mark.parametrize("rows,M", SHAPES)
def test_mlp_fused_bwd(ops, rows, M):
    x = rnd("mlpb/x", (rows, D), 1.5)
    ln_w, ln_b, w1, b1, w2, b2 = params("mlpb", M)
    dy = rnd("mlpb/dy", (rows, D), 1.0)
    out, h, mean, rstd, gd, _ = ops.mlp_fwd(x, ln_w, ln_b, w1.to(_H.td), b1, w2.to(_H.td), b2, _H.name)
    w2t = w2.to(_H.td).T.contiguous()                         # (M, D)
    w1t = w1.to(_H.td).T.contiguous()                         # (D, M)
    dx, dx_c, du, partials = ops.mlp_bwd(dy, dy.to(_H.td), x, mean, rstd, ln_w, w2t, w1t, gd, _H.name)
    torch.cuda.synchronize()
    du_r = (r16(dy) @ r16(w2)) * (gd.double() + 0.5)           # the saved (16-bit, centred) derivative is what backward sees
    assert rel(du, du_r) < 5e-3
    u_r = h.double() @ r16(w1).T + b1.double()                 # ... and against the exact derivative of the exact pre-activation
    assert rel(du, (r16(dy) @ r16(w2)) * gelu_grad(u_r)) < 7e-3
    # LayerNorm backward from the kernel's own du (bf16): autograd in float64
    xd = x.double().requires_grad_(True)
    lw = ln_w.double().requires_grad_(True)
    lb = ln_b.double().requires_grad_(True)
    hd = torch.nn.functional.layer_norm(xd, (D,), lw, lb, 1e-5)
    hd.backward(du.double() @ r16(w1))
    dx_r = dy.double() + xd.grad
    assert rel(dx - dy, xd.grad) < 2e-3
    assert rel(dx, dx_r) < 1e-3
    assert rel(dx_c, dx_r) < 4e-3
    # rows per workgroup (fused_block_rows in csrc/fused_epilogue.h: fewer rounds of 256 workgroups x rows wins, ties to 96)
    blk = 96 if ((rows + 95) // 96 + 255) // 256 * 96 <= ((rows + 127) // 128 + 255) // 256 * 128 else 128
    assert partials.shape == ((rows + blk - 1) // blk, 2, D)
    assert rel(partials[:, 0].sum(0), lw.grad) < 2e-3
    assert rel(partials[:, 1].sum(0), lb.grad) < 2e-3


def test_mlp_fused_bwd_zero_gradient_rows(ops):
    """dy = 0 on some rows (the cls-only gradient of pool='cls' is zero almost everywhere at the top
    layer): those rows must produce exactly dx = 0 and contribute nothing to dgamma / dbeta."""
    rows, M = 300, 768
    x = rnd("mlpz/x", (rows, D), 1.5)
    ln_w, ln_b, w1, b1, w2, b2 = params("mlpz", M)
    dy = rnd("mlpz/dy", (rows, D), 1.0)
    dy[::3] = 0
    out, h, mean, rstd, gd, _ = ops.mlp_fwd(x, ln_w, ln_b, w1.to(_H.td), b1, w2.to(_H.td), b2, _H.name)
    dx, dx_c, du, partials = ops.mlp_bwd(dy, dy.to(_H.td), x, mean, rstd, ln_w, w2.to(_H.td).T.contiguous(),
                                         w1.to(_H.td).T.contiguous(), gd, _H.name)
    assert float(dx[::3].abs().max()) == 0.0 and float(du[::3].float().abs().max()) == 0.0
    assert bool(torch.isfinite(dx).all()) and bool(torch.isfinite(partials).all())


def test_mlp_fused_repeatable_and_in_bounds(ops):
    """The kernels keep asynchronous LDS reads and stores in flight behind hand-placed waits: repeated
    launches at the full BASELINE shape must give identical bits (a missing wait shows as lane-pattern
    noise), and rows past R in the ragged last workgroup must not be written (sentinel rows stay intact)."""
    from sitk import runtime as rt
    rows, M, pad = 20544 - 37, 768, 128
    x = rnd("mlpr/x", (rows, D), 1.5)
    ln_w, ln_b, w1, b1, w2, b2 = params("mlpr", M)
    dy = rnd("mlpr/dy", (rows, D), 1.0)
    w1c, w2c = w1.to(_H.td), w2.to(_H.td)
    w1t, w2t = w1c.T.contiguous(), w2c.T.contiguous()
    ref = None
    for _ in range(4):
        out, h, mean, rstd, u, g = ops.mlp_fwd(x, ln_w, ln_b, w1c, b1, w2c, b2, _H.name, want_g=True)
        got = (out, h, u, g) + ops.mlp_bwd(dy, dy.to(_H.td), x, mean, rstd, ln_w, w2t, w1t, u, _H.name)
        torch.cuda.synchronize()
        if ref is None:
            ref = [t.clone() for t in got]
        else:
            for a, b in zip(ref, got):
                assert torch.equal(a, b)
    # sentinel rows behind every row-indexed output of the forward kernel
    big = lambda cols, dt: torch.full((rows + pad, cols), 7.0, dtype=dt, device=DEV)  # noqa: E731
    out_b, h_b, u_b, g_b = big(D, torch.float32), big(D, _H.td), big(M, _H.td), big(M, _H.td)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    rt.check(rt.lib.sitk_mlp_fwd(x.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), w1c.data_ptr(), b1.data_ptr(), w2c.data_ptr(),
                                 b2.data_ptr(), h_b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), u_b.data_ptr(), g_b.data_ptr(),
                                 out_b.data_ptr(), rows, D, M, rt.dtype_code(_H.name), rt.stream_ptr()))
    torch.cuda.synchronize()
    for t, r in ((out_b, ref[0]), (h_b, ref[1]), (u_b, ref[2]), (g_b, ref[3])):
        assert torch.equal(t[:rows], r)
        assert bool((t[rows:].float() == 7.0).all())


@pytest.mark.parametrize("rows", [96, 321 * 3, 1000, 20544])
def test_attn_out_mlp_fwd(ops, rows):
    """to_out + residual folded into the fused forward: x_mid = x + o Wo^T + bo, out = x_mid + MLP(LN(x_mid)).
    Checked against float64 torch math on bf16-rounded operands, and the MLP part against the stand-alone fused
    kernel run on the kernel's own x_mid (identical bits: same code path after the prologue)."""
    M = 768
    assert ops.attn_out_mlp_fused_supported(rows, D, 192, M, _H.name)
    assert not ops.attn_out_mlp_fused_supported(30000, D, 192, M, _H.name)      # 128-row workgroups: separate kernels
    assert not ops.attn_out_mlp_fused_supported(rows, D, 384, M, _H.name)
    x = rnd("aom/x", (rows, D), 1.5)
    o = rnd("aom/o", (rows, D), 1.0).to(_H.td)
    wo, bo = rnd("aom/wo", (D, D), D ** -0.5), rnd("aom/bo", (D,), 0.1)
    ln_w, ln_b, w1, b1, w2, b2 = params("aom", M)
    out, xmid, h, mean, rstd, u, g = ops.attn_out_mlp_fwd(o, wo.to(_H.td), bo, x, ln_w, ln_b, w1.to(_H.td), b1,
                                                          w2.to(_H.td), b2, _H.name, want_g=True)
    torch.cuda.synchronize()
    xm_r = x.double() + o.double() @ r16(wo).T + bo.double()
    assert rel(xmid, xm_r) < 2e-6                              # fp32 accumulation of bf16 x bf16 products
    assert rel(xmid - x, xm_r - x.double()) < 2e-5
    ref = ops.mlp_fwd(xmid, ln_w, ln_b, w1.to(_H.td), b1, w2.to(_H.td), b2, _H.name, want_g=True)
    for a, b in zip((out, h, mean, rstd, u, g), ref):
        assert torch.equal(a, b)
    # sentinel rows behind x_mid
    from sitk import runtime as rt
    pad = 128
    xm_b = torch.full((rows + pad, D), 7.0, device=DEV)
    out_b = torch.empty((rows, D), device=DEV)
    woc, w1c, w2c = wo.to(_H.td), w1.to(_H.td), w2.to(_H.td)          # keep the operands alive across the raw call
    rt.check(rt.lib.sitk_attn_out_mlp_fwd(o.data_ptr(), woc.data_ptr(), bo.data_ptr(), x.data_ptr(), xm_b.data_ptr(),
                                          ln_w.data_ptr(), ln_b.data_ptr(), w1c.data_ptr(), b1.data_ptr(),
                                          w2c.data_ptr(), b2.data_ptr(), 0, 0, 0, 0, 0, out_b.data_ptr(), rows, D, 192,
                                          M, rt.dtype_code(_H.name), rt.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(xm_b[:rows], xmid) and bool((xm_b[rows:] == 7.0).all()) and torch.equal(out_b, out)


@pytest.mark.parametrize("rows", [96, 321 * 3, 1000, 20544])
def test_attn_out_mlp_next_fwd(ops, rows):
    """... + the next block's LayerNorm and to_qkv appended: every output must carry the bits of the separate
    launches (attn_out_mlp_fwd, then ln_gemm_fwd on its `out`)."""
    M, N3 = 768, 576
    x = rnd("aon/x", (rows, D), 1.5)
    o = rnd("aon/o", (rows, D), 1.0).to(_H.td)
    woc, bo = rnd("aon/wo", (D, D), D ** -0.5).to(_H.td), rnd("aon/bo", (D,), 0.1)
    ln_w, ln_b, w1, b1, w2, b2 = params("aon", M)
    w1c, w2c = w1.to(_H.td), w2.to(_H.td)
    n_lw, n_lb = rnd("aon/nlw", (D,), 0.3) + 1.0, rnd("aon/nlb", (D,), 0.2)
    wq = rnd("aon/wq", (N3, D), D ** -0.5).to(_H.td)
    got = ops.attn_out_mlp_next_fwd(o, woc, bo, x, ln_w, ln_b, w1c, b1, w2c, b2, n_lw, n_lb, wq, _H.name, want_g=True)
    ref = ops.attn_out_mlp_fwd(o, woc, bo, x, ln_w, ln_b, w1c, b1, w2c, b2, _H.name, want_g=True)
    qkv, nh, nmean, nrstd = ops.ln_gemm_fwd(ref[0], n_lw, n_lb, wq, _H.name)
    torch.cuda.synchronize()
    for a, b in zip(got[:7], ref):
        assert torch.equal(a, b)
    assert torch.equal(got[7], nh) and torch.equal(got[10], qkv)
    assert rel(got[8], nmean) < 1e-6 and rel(got[9], nrstd) < 1e-6
    # repeatable
    again = ops.attn_out_mlp_next_fwd(o, woc, bo, x, ln_w, ln_b, w1c, b1, w2c, b2, n_lw, n_lb, wq, _H.name, want_g=True)
    for a, b in zip(got, again):
        assert torch.equal(a, b)


@pytest.mark.parametrize("rows", [96, 321 * 3, 1000, 20544])
def test_ln_gemm_mlp_bwd_pair_launch_is_bitwise_the_two_launches(ops, rows):
    """sitk_ln_gemm_mlp_bwd = sitk_ln_gemm_bwd (d to_qkv + LayerNorm backward of layer l) followed by sitk_mlp_bwd of layer
    l - 1 on the dx / dx_c it wrote, in ONE launch (same 96-row workgroups; the second half reads its own rows back): every
    output must carry the bits of the two separate launches, repeatedly (a missing wait between the halves shows as noise)."""
    M, N3 = 768, 576
    assert ops.ln_gemm_mlp_bwd_supported(rows, D, N3, M, _H.name)
    assert not ops.ln_gemm_mlp_bwd_supported(30000, D, N3, M, _H.name)           # 128-row workgroups: separate launches
    x, xmid = rnd("pair/x", (rows, D), 1.5), rnd("pair/xm", (rows, D), 1.5)
    dres = rnd("pair/dres", (rows, D), 1.0)
    dqkv = rnd("pair/dqkv", (rows, N3), 1.0).to(_H.td)
    wq_t = rnd("pair/wq", (D, N3), D ** -0.5).to(_H.td)
    ln1_w = rnd("pair/l1", (D,), 0.3) + 1.0
    ln_w, ln_b, w1, b1, w2, b2 = params("pair", M)
    w1c, w2c = w1.to(_H.td), w2.to(_H.td)
    w1t, w2t = w1c.T.contiguous(), w2c.T.contiguous()
    _, _, mean2, rstd2, gd, _ = ops.mlp_fwd(xmid, ln_w, ln_b, w1c, b1, w2c, b2, _H.name)
    xd = x.double()
    mean1, rstd1 = xd.mean(1).float(), (xd.var(1, unbiased=False) + 1e-5).rsqrt().float()
    dx, dx_c, p1 = ops.ln_gemm_bwd(dqkv, wq_t, x, mean1, rstd1, ln1_w, dres, _H.name)
    dxm, dxm_c, du, p2 = ops.mlp_bwd(dx, dx_c, xmid, mean2, rstd2, ln_w, w2t, w1t, gd, _H.name)
    torch.cuda.synchronize()
    for _ in range(3):
        got = ops.ln_gemm_mlp_bwd(dqkv, wq_t, x, mean1, rstd1, ln1_w, dres, xmid, mean2, rstd2, ln_w, w2t, w1t, gd, _H.name)
        torch.cuda.synchronize()
        for a, b in zip(got, (dx, dx_c, p1, dxm, dxm_c, du, p2)):
            assert torch.equal(a, b)


@pytest.mark.parametrize("rows", [96, 1000, 20544, 30000])
def test_mlp_bwd_that_rounds_dy_itself_is_bitwise_cast_plus_mlp_bwd(ops, rows):
    """sitk_mlp_bwd_cast (the first backward launch of a chain: dy exists in fp32 only) = sitk_cast_rows + sitk_mlp_bwd, bit for
    bit, and the compute-dtype copy it writes for the weight gradients is the cast's."""
    M = 768
    xmid = rnd("bc/xm", (rows, D), 1.5)
    dy = rnd("bc/dy", (rows, D), 1.0)
    ln_w, ln_b, w1, b1, w2, b2 = params("bc", M)
    w1c, w2c = w1.to(_H.td), w2.to(_H.td)
    w1t, w2t = w1c.T.contiguous(), w2c.T.contiguous()
    _, _, mean2, rstd2, gd, _ = ops.mlp_fwd(xmid, ln_w, ln_b, w1c, b1, w2c, b2, _H.name)
    dy_c = ops.cast_rows(dy, _H.name)
    want = ops.mlp_bwd(dy, dy_c, xmid, mean2, rstd2, ln_w, w2t, w1t, gd, _H.name)
    got = ops.mlp_bwd(dy, None, xmid, mean2, rstd2, ln_w, w2t, w1t, gd, _H.name)
    torch.cuda.synchronize()
    for a, b in zip(got[:4], want):
        assert torch.equal(a, b)
    assert torch.equal(got[4], dy_c)
