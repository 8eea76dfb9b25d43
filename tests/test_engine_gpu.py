"""GPU tests of the fused train step (sitk.engine.TrainEngine): one step == autograd modules +
torch.optim.SGD on the same batch; hipGraph replay == eager; MPP engine gradients == autograd path
replaying the engine's own random draws."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detgen, sit_oracle  # noqa: E402

from tests.parity_bars import check  # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _load(module, seed):
    vals = detgen.fill_state_dict(module.state_dict(), seed=seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})


@pytest.fixture(scope="module")
def pk():
    import sitk  # noqa: F401
    from sitk import engine
    from sitk.models import mpp, sit
    return sit, mpp, engine


@pytest.mark.parametrize("dtype,layout,pool", [("f32", "surface", "cls"), ("bf16", "patched", "mean"), ("bf16", "surface", "cls"),
                                               ("f16", "patched", "mean"), ("f16", "surface", "cls")])
def test_engine_step_equals_autograd_plus_sgd(pk, dtype, layout, pool):
    sit, _, engine = pk
    B = 4
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], depth=2, num_patches=320, num_vertices=153, num_channels=4, pool=pool)
    m1 = sit.SiT(**kw, compute_dtype=dtype)
    _load(m1, 11)
    m2 = copy.deepcopy(m1)
    m1.to(DEV)
    if layout == "surface":
        x = torch.from_numpy(detgen.normal("en/xs", (B, 40962, 4), seed=1)).to(DEV)
    else:
        x = torch.from_numpy(detgen.normal("en/xp", (B, 4, 320, 153), seed=1)).to(DEV)
    y = torch.from_numpy(detgen.normal("en/y", (B,), seed=1)).to(DEV)
    lr = 0.05
    opt = torch.optim.SGD(m1.parameters(), lr=lr, momentum=0.9)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(m1(x).squeeze(), y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    for use_graph in (False, True):
        m = copy.deepcopy(m2)
        eng = engine.TrainEngine(m, B, input_layout=layout, lr=lr, momentum=0.9, use_graph=use_graph)
        got = [float(eng.step(x, y)) for _ in range(3)]
        case = f"engine/{layout}_{pool}_{'graph' if use_graph else 'eager'}"
        check(case, "loss", dtype, max(abs(a - b) / abs(b) for a, b in zip(got, losses)), "out")
        # the two bf16 paths round differently (bf16 vs fp32 dY operands of the weight gradients)
        worst = max((rel(p.data, q.data), k) for (k, p), (_, q) in zip(m.named_parameters(), m1.named_parameters()))
        print("worst parameter:", worst)
        check(case, "param", dtype, worst[0], "param")
        assert eng.fp.still_flat()
        sd = m.state_dict()
        assert rel(sd["pos_embedding"], m1.state_dict()["pos_embedding"]) < 1e-3


def test_engine_backward_slices_match_single_slice(pk):
    sit, _, engine = pk
    B = 2
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], depth=4, num_patches=80, num_vertices=561, num_channels=4)
    base = sit.SiT(**kw, compute_dtype="f32")
    _load(base, 5)
    x = torch.from_numpy(detgen.normal("sl/x", (B, 4, 80, 561), seed=1)).to(DEV)
    y = torch.from_numpy(detgen.normal("sl/y", (B,), seed=1)).to(DEV)
    flats = []
    for slices in (1, 3):
        eng = engine.TrainEngine(copy.deepcopy(base), B, input_layout="patched", lr=0.1, bwd_slices=slices, use_graph=False)
        eng.step(x, y)
        flats.append(eng.fp.flat.clone())
    assert rel(flats[1], flats[0]) < 1e-6


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_mpp_engine_gradients_match_autograd_path(pk, dtype):
    sit, mpp, engine = pk
    B, P, V = 3, 320, 153
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], depth=2, num_patches=P, num_vertices=V, num_channels=4)
    ssl = mpp.masked_patch_pretraining(sit.SiT(**kw, compute_dtype=dtype), 192, 4 * V, "cpu", mask_prob=0.75,
                                       replace_prob=0.8, swap_prob=0.02, channels=4, num_vertices=V)
    _load(ssl, 5)
    ref = copy.deepcopy(ssl).to(DEV)
    x = torch.from_numpy(detgen.normal("me/x", (B, 40962, 4), seed=1)).to(DEV)
    eng = engine.TrainEngine(ssl, B, task="mpp", input_layout="surface", lr=0.0, momentum=0.0, use_graph=False, keep_grads=True)
    torch.manual_seed(0)
    loss = float(eng.step(x))
    rnd = {k: v.clone() for k, v in eng.last_randoms.items()}
    assert int(rnd["corrupted_sequence"].sum()) == B * 240
    l2, _ = ref(x, randoms=rnd)
    l2.backward()
    check("engine/mpp_tiny320", "loss", dtype, abs(loss - float(l2)) / float(l2), "out")
    worst = (0.0, "")
    for (k, p), (_, q) in zip(ssl.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            assert float(p.grad.abs().max()) == 0.0, k
            continue
        worst = max(worst, (rel(p.grad, q.grad), k))
    print("worst gradient:", worst)
    check("engine/mpp_tiny320", "grad_rel", dtype, worst[0], "grad")
    # graph-captured MPP steps run and reduce the loss
    ssl2 = copy.deepcopy(ref).cpu()
    eng2 = engine.TrainEngine(ssl2, B, task="mpp", input_layout="surface", lr=0.02, momentum=0.9, use_graph=True)
    ls = [float(eng2.step(x)) for _ in range(12)]
    assert all(np.isfinite(ls)) and np.mean(ls[-3:]) < np.mean(ls[:3]), ls


@pytest.mark.parametrize("dtype,optimizer", [("f32", "sgd"), ("bf16", "sgd"), ("f32", "adamw")])
def test_mpp_engine_optimizer_scope_of_the_reference_loop(pk, dtype, optimizer):
    """tools/pretrain.py:267-280 builds its optimizer over `model.parameters()` -- the SiT -- so
    `to_original.*` and `mask_token` (models/mpp.py:66,74) keep their initial values for the whole run, and `mlp_head.*` (not on
    the MPP path: grad None) is skipped by torch's optimizers, weight decay and all.  TrainEngine(task='mpp', optimize='sit')
    is that loop: three steps with weight decay > 0 against the module path + torch.optim.X(model.parameters()) replaying the
    engine's draws -- the SiT's parameters follow, `to_original.*`, `mask_token`, `mlp_head.*` stay BIT-unchanged.
    optimize='all' (the default) updates the MPP head too (and still leaves mlp_head alone)."""
    sit, mpp, engine = pk
    B, P, V = 3, 320, 153
    lr, wd = 0.02, 0.05
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], depth=2, num_patches=P, num_vertices=V, num_channels=4)
    ssl = mpp.masked_patch_pretraining(sit.SiT(**kw, compute_dtype=dtype), 192, 4 * V, "cpu", mask_prob=0.75,
                                       replace_prob=0.8, swap_prob=0.02, channels=4, num_vertices=V)
    _load(ssl, 5)
    init = {k: p.detach().clone() for k, p in ssl.named_parameters()}
    ref = copy.deepcopy(ssl).to(DEV)
    ssl_all = copy.deepcopy(ssl)
    x = torch.from_numpy(detgen.normal("me/x", (B, 40962, 4), seed=1)).to(DEV)
    if optimizer == "sgd":
        opt = torch.optim.SGD(ref.transformer.parameters(), lr=lr, momentum=0.9, weight_decay=wd)
        ekw = dict(optimizer="sgd", momentum=0.9, weight_decay=wd)
    else:
        opt = torch.optim.AdamW(ref.transformer.parameters(), lr=lr, weight_decay=wd)
        ekw = dict(optimizer="adamw", weight_decay=wd)
    eng = engine.TrainEngine(ssl, B, task="mpp", input_layout="surface", lr=lr, use_graph=False, optimize="sit", **ekw)
    assert eng.n_opt < eng.fp.total
    lerr = 0.0
    for _ in range(3):
        l_eng = float(eng.step(x))
        rnd = {k: v.clone() for k, v in eng.last_randoms.items()}
        opt.zero_grad()
        l_ref, _ = ref(x, randoms=rnd)
        l_ref.backward()
        opt.step()
        lerr = max(lerr, abs(l_eng - float(l_ref)) / float(l_ref))
    check(f"engine/mpp_sit_scope_{optimizer}", "loss", dtype, lerr, "out")
    frozen = ("to_original.", "mask_token", "transformer.mlp_head.")
    worst = (0.0, "")
    for (k, p), (_, q) in zip(ssl.named_parameters(), ref.named_parameters()):
        if k.startswith(frozen):
            assert torch.equal(p.detach().cpu(), init[k]), f"{k} moved under optimize='sit'"
            assert torch.equal(q.detach().cpu(), init[k]), f"{k} moved on the reference-style module path"
            continue
        upd = q.detach().cpu() - init[k]
        assert float(upd.abs().max()) > 0, k
        worst = max(worst, (rel(p.detach().cpu() - init[k], upd), k))
    print("worst parameter update:", worst)
    if optimizer == "adamw":
        # AdamW divides by sqrt(v): an element whose gradient is ~0 moves by +- lr on the sign of rounding noise, and the f32 MPP
        # path still has two order-dependent sums (section 2 of DESIGN.md) -- two runs measured 6.8e-4 and 1.5e-3 of the update
        # on the worst tensor.  What this case pins is the SCOPE (above: bit-unchanged tensors); the update itself to 1e-2.
        assert worst[0] < 1e-2, worst
    else:
        check(f"engine/mpp_sit_scope_{optimizer}", "update_rel", dtype, worst[0], "grad")
    assert float(eng.fp.grad_all.abs().max()) == 0.0              # the frozen parameters' gradients are cleared with the rest
    # optimize='all': the head of models/mpp.py moves, mlp_head still does not
    eng_all = engine.TrainEngine(ssl_all, B, task="mpp", input_layout="surface", lr=lr, use_graph=False, **ekw)
    assert eng_all.optimize == "all"
    eng_all.step(x)
    for k, p in ssl_all.named_parameters():
        moved = not torch.equal(p.detach().cpu(), init[k])
        assert moved == (not k.startswith("transformer.mlp_head.")), (k, moved)


@pytest.mark.parametrize("mode", ["graph", "side_stream"])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_engine_bench_config_matches_autograd_path(pk, dtype, mode):
    """BASELINE config 2 end to end, exactly as bench.py runs it: SiT-tiny, depth 12, B = 64, bf16, raw surfaces,
    one hipGraph per segment -- two steps against the autograd module path + torch.optim.SGD on the same batch
    (tools/train.py:280-291).  Compared: both losses and the parameter UPDATE (after - before) of every tensor."""
    sit, _, engine = pk
    B, lr = 64, 0.01
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], num_patches=320, num_vertices=153, num_channels=4)
    m1 = sit.SiT(**kw, compute_dtype=dtype)
    _load(m1, 21)
    m2 = copy.deepcopy(m1)
    before = {k: p.detach().clone() for k, p in m1.named_parameters()}
    m1.to(DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((B, 40962, 4), device=DEV, generator=g)
    y = torch.randn((B,), device=DEV, generator=g) * 2 + 40
    opt = torch.optim.SGD(m1.parameters(), lr=lr, momentum=0.9)
    losses = []
    for _ in range(2):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(m1(x).squeeze(), y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    # "graph": one hipGraph per step; "side_stream": bench.py's default form -- eager launches, the weight gradients of the
    # first 7 finished layers on the engine's side stream beside the rest of backward (sitk_encoder_bwd_overlap)
    eng = engine.TrainEngine(m2, B, input_layout="surface", lr=lr, momentum=0.9, use_graph=True if mode == "graph" else None)
    got = [float(eng.step(x, y)) for _ in range(2)]
    if mode == "graph":
        assert eng._graphs and not eng._overlap, "the step must have been captured"
    else:
        assert eng._overlap and not eng.use_graph, "the default form of this configuration forks the side stream"
    check("engine/bench_tiny_b64", "loss", dtype, max(abs(a - b) / abs(b) for a, b in zip(got, losses)), "out")
    worst = (0.0, "")
    for (k, p), (_, q) in zip(m2.named_parameters(), m1.named_parameters()):
        d_eng, d_ref = p.detach().cpu() - before[k], q.detach().cpu() - before[k]
        worst = max(worst, (rel(d_eng, d_ref), k))
    print("worst parameter update:", worst)
    check("engine/bench_tiny_b64", "update_rel", dtype, worst[0], "grad")
    assert eng.fp.still_flat()


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_engine_bench_config_against_cpu_oracle(pk, dtype):
    """VERDICT r4 next 2(a) / ADVICE: the BENCHMARKED configuration itself -- SiT-tiny, depth 12, 320 patches, B = 64, raw
    (B, 40962, 4) surfaces, the engine's DEFAULT launch form (16-bit: eager, 8 layers' weight gradients on the side stream, the
    chained `d to_qkv` + MLP backward and merged attention backward launches, prefetched gather) -- ONE step with kept
    gradients against oracle/sit_oracle.py on the CPU on the same batch: the reference's loop body, tools/train.py:280-291
    (forward, MSE, backward).  Until round 5 this configuration met the oracle at B = 4 only (golden tiny320_cls) and the engine
    at B = 64 only another HIP path.  Bars: f32 2e-4 / 1e-3; f16 north_star's fixed 1e-3 on the loss, on every gradient's NORM
    and on every gradient element-wise (relative to the tensor); bf16 under its recorded bars and fixed ceilings."""
    sit, _, engine = pk
    B = 64
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], num_patches=320, num_vertices=153, num_channels=4)
    m = sit.SiT(**kw, compute_dtype=dtype)
    _load(m, 21)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((B, 40962, 4), generator=g)
    y = torch.randn((B,), generator=g) * 2 + 40
    ref = sit_oracle.SiT(**kw)
    ref.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
    from sitk import tables
    table = tables.load_table(320, 153)                                                 # (P, V) uint16, sha-pinned to the reference CSV
    tok = torch.from_numpy(sit_oracle.gather_tokens(x.numpy(), table))                  # (B, P, V * C), f = v * C + c
    xp = tok.reshape(B, 320, 153, 4).permute(0, 3, 1, 2).contiguous()                   # the reference's (B, C, P, V) input
    l_ref, g_ref = _oracle_grads_cpu(ref, lambda mod: torch.nn.functional.mse_loss(mod(xp).squeeze(-1), y))
    eng = engine.TrainEngine(m, B, input_layout="surface", lr=0.0, momentum=0.0, keep_grads=True)
    if dtype != "f32":
        assert eng._overlap and not eng.use_graph and eng.wgrad_overlap == 8 and eng._prefetch, "not the benchmarked launch form"
    loss = float(eng.step(x.to(DEV), y.to(DEV)))
    case = "engine/bench_tiny_b64_oracle"
    check(case, "loss", dtype, abs(loss - l_ref) / abs(l_ref), "loss")
    worst_n, worst_e = (0.0, ""), (0.0, "")
    for k, p in m.named_parameters():
        gn, rn = float(p.grad.double().norm()), float(g_ref[k].double().norm())
        worst_n = max(worst_n, (abs(gn - rn) / rn, k))
        worst_e = max(worst_e, (rel(p.grad, g_ref[k]), k))
    print("worst gradient norm:", worst_n, " worst gradient (element-wise, relative to the tensor):", worst_e)
    check(case, "gnorm", dtype, worst_n[0], "grad")
    check(case, "grad_rel", dtype, worst_e[0], "grad")
    assert eng.nonfinite_count == 0


@pytest.mark.parametrize("mode", ["graph", "side_stream"])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_engine_bench_config_is_bitwise_reproducible(pk, dtype, mode):
    """no float atomics on the bench path any more (the head's parameter gradients and the loss go
    through per-sample partial rows, d pos_embedding / d cls_token and the LayerNorm gradients through ordered sums, the
    weight gradients of the one-launch 12-layer slice write each tile once).  Two engines from the same weights, the
    bench configuration (tiny, depth 12, B = 64, bf16, raw surfaces, hipGraph), three steps: every float of the flat
    parameter buffer and every loss must be BIT-equal."""
    sit, _, engine = pk
    B = 64
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], num_patches=320, num_vertices=153, num_channels=4)
    base = sit.SiT(**kw, compute_dtype=dtype)
    _load(base, 21)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((B, 40962, 4), device=DEV, generator=g)
    y = torch.randn((B,), device=DEV, generator=g) * 2 + 40
    runs = []
    for _ in range(2):
        eng = engine.TrainEngine(copy.deepcopy(base), B, input_layout="surface", lr=0.01, momentum=0.9,
                                 use_graph=True if mode == "graph" else None)
        assert bool(eng._overlap) == (mode == "side_stream")
        losses = [eng.step(x, y).clone() for _ in range(3)]
        torch.cuda.synchronize()
        runs.append((eng.fp.flat.clone(), torch.cat(losses)))
    assert torch.equal(runs[0][1], runs[1][1]), (runs[0][1], runs[1][1])
    diff = int((runs[0][0] != runs[1][0]).sum())
    assert diff == 0, f"{diff} of {runs[0][0].numel()} parameters differ between two identical runs"


def _oracle_grads_cpu(model_cpu, fwd):
    torch.set_num_threads(max(1, min(16, (torch.get_num_threads() or 1))))
    model_cpu.zero_grad()
    loss = fwd(model_cpu)
    loss.backward()
    return float(loss), {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in model_cpu.named_parameters()}


def test_engine_config3_full_width_f32_against_oracle(pk):
    """VERDICT r2 weak #4 / ADVICE: BASELINE config 3 at its FULL width -- SiT-small, 1280 patches (N = 1281), B = 32 --
    depth 1, f32 compute mode, against the CPU oracle on the same batch (the goldens pin these kernels at B <= 2 only; the
    bf16-vs-f32-mode test below is a self-comparison).  Loss <= 2e-4, every gradient <= 1e-3 (north_star's bar)."""
    sit, _, engine = pk
    B = 32
    kw = dict(sit_oracle.MODEL_SIZES["small"], num_patches=1280, num_vertices=45, num_channels=4)
    kw["depth"] = 1
    g = torch.Generator().manual_seed(11)
    x = torch.randn((B, 4, 1280, 45), generator=g)
    y = torch.randn((B,), generator=g) * 2 + 40
    m = sit.SiT(**kw, compute_dtype="f32")
    _load(m, 33)
    ref = sit_oracle.SiT(**kw)
    ref.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
    l_ref, g_ref = _oracle_grads_cpu(ref, lambda mod: torch.nn.functional.mse_loss(mod(x).squeeze(-1), y))
    eng = engine.TrainEngine(m, B, input_layout="patched", lr=0.0, momentum=0.0, use_graph=False, keep_grads=True)
    loss = float(eng.step(x.to(DEV), y.to(DEV)))
    check("engine/cfg3_b32_d1_oracle", "loss", "f32", abs(loss - l_ref) / abs(l_ref), "loss")
    worst = max((rel(p.grad, g_ref[k]), k) for k, p in m.named_parameters())
    print("worst gradient:", worst)
    check("engine/cfg3_b32_d1_oracle", "grad_rel", "f32", worst[0], "grad")


def test_mpp_engine_config5_full_width_f32_against_oracle(pk):
    """BASELINE config 5's per-GPU share at FULL width -- SiT-base MPP, 1280 patches x 45 vertices, 32 samples -- depth 1,
    f32 compute mode: the engine's own device draws replayed through the CPU oracle (models/mpp.py:77-134 restated in
    oracle/sit_oracle.py).  Loss <= 2e-4, every gradient <= 1e-3."""
    sit, mpp, engine = pk
    B, P, V = 32, 1280, 45
    kw = dict(sit_oracle.MODEL_SIZES["base"], depth=1, num_patches=P, num_vertices=V, num_channels=4)
    g = torch.Generator().manual_seed(3)
    x = torch.randn((B, 4, P, V), generator=g)
    model = sit.SiT(**kw, compute_dtype="f32")
    model.allow_synthetic_table = True
    ssl = mpp.masked_patch_pretraining(model, 768, 4 * V, "cpu", mask_prob=0.75, replace_prob=0.8, swap_prob=0.02,
                                       channels=4, num_vertices=V)
    _load(ssl, 9)
    ref = sit_oracle.MaskedPatchPretraining(sit_oracle.SiT(**kw), 768, 4 * V, "cpu", mask_prob=0.75, replace_prob=0.8,
                                            swap_prob=0.02, channels=4, num_vertices=V)
    ref.load_state_dict({k: v.detach().clone() for k, v in ssl.state_dict().items()})
    eng = engine.TrainEngine(ssl, B, task="mpp", input_layout="patched", lr=0.0, momentum=0.0, use_graph=False,
                             keep_grads=True)
    loss = float(eng.step(x.to(DEV)))
    rnd = {k: v.cpu() for k, v in eng.last_randoms.items()}
    l_ref, g_ref = _oracle_grads_cpu(ref, lambda mod: mod(x, randoms=rnd)[0])
    check("engine/cfg5_mpp_b32_d1_oracle", "loss", "f32", abs(loss - l_ref) / abs(l_ref), "loss")
    worst = (0.0, "")
    for k, p in ssl.named_parameters():
        if g_ref[k] is None:
            assert float(p.grad.abs().max()) == 0.0, k
            continue
        worst = max(worst, (rel(p.grad, g_ref[k]), k))
    print("worst gradient:", worst)
    check("engine/cfg5_mpp_b32_d1_oracle", "grad_rel", "f32", worst[0], "grad")


_ORACLE_CACHE = {}


def _oracle_grads_cpu_chunked(key, model_cpu, n, chunk, loss_of_chunk):
    """Loss and gradients of the CPU oracle over n samples in chunks of `chunk` (a depth-12 model on 1 281 tokens keeps
    (chunk, heads, N, N) softmax outputs per layer for its backward: 32 samples at once are tens of GB).  loss_of_chunk(model,
    lo, hi) returns that chunk's share of the batch loss (its mean times (hi - lo) / n), so the sum is the batch loss and the
    accumulated .grad its gradient.  Cached per `key`: the two 16-bit modes of one test compare with the same oracle run."""
    if key in _ORACLE_CACHE:
        return _ORACLE_CACHE[key]
    torch.set_num_threads(max(1, min(16, (torch.get_num_threads() or 1))))
    model_cpu.zero_grad()
    total = 0.0
    for lo in range(0, n, chunk):
        part = loss_of_chunk(model_cpu, lo, min(lo + chunk, n))
        part.backward()
        total += float(part.detach())
    out = total, {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in model_cpu.named_parameters()}
    _ORACLE_CACHE.clear()                     # one entry at a time: base-width gradients are 350 MB
    _ORACLE_CACHE[key] = out
    return out


def _worst_grads(named_parameters, g_ref):
    worst_n, worst_e = (0.0, ""), (0.0, "")
    for k, p in named_parameters:
        if g_ref[k] is None:
            assert float(p.grad.abs().max()) == 0.0, k
            continue
        gn, rn = float(p.grad.double().norm()), float(g_ref[k].double().norm())
        worst_n = max(worst_n, (abs(gn - rn) / rn, k))
        worst_e = max(worst_e, (rel(p.grad, g_ref[k]), k))
    return worst_n, worst_e


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_engine_config3_benchmarked_form_against_cpu_oracle(pk, dtype):
    """the engine form `bench.py --model small --patches 1280 --batch 32` times (`also.cfg3`, BASELINE
    config 3) -- SiT-small, 1280 patches x 45 vertices (the synthetic table), DEPTH 12, B = 32, raw (B, 40962, 4) surfaces,
    the launch form bench.py times for this width (engine.make_engine: two concurrent half-batch steps on two streams, each ONE
    hipGraph -- the stand-alone LayerNorms, the 2-per-CU N % 192 GEMMs, ring attention on 1 281 tokens, the one-launch weight
    gradients -- and one optimizer pass over both gradient buffers) -- one REPLAYED step with
    kept gradients against oracle/sit_oracle.py on the same batch (the reference's loop body, tools/train.py:280-291 on
    config/SiT/training/hparams.yml:34's shapes).  Until round 6 the oracle met this engine form at depth 1 in f32 only; the
    depth-12 golden `small1280_d12` goes through the autograd module path at B = 2.  f16: north_star's fixed 1e-3 on the
    loss, every gradient's norm and every gradient element-wise; bf16: recorded bars under the fixed ceilings.  CPU side:
    8.1 TFLOP in chunks of 8 samples (~20 s on the box's 16 cores)."""
    sit, _, engine = pk
    B, P, V = 32, 1280, 45
    kw = dict(sit_oracle.MODEL_SIZES["small"], num_patches=P, num_vertices=V, num_channels=4)
    assert kw["depth"] == 12
    m = sit.SiT(**kw, compute_dtype=dtype)
    m.allow_synthetic_table = True
    _load(m, 33)
    g = torch.Generator().manual_seed(17)
    x = torch.randn((B, 40962, 4), generator=g)
    y = torch.randn((B,), generator=g) * 2 + 40
    from sitk import tables
    table = tables.load_table(P, V, allow_synthetic=True)
    tok = torch.from_numpy(sit_oracle.gather_tokens(x.numpy(), table))
    xp = tok.reshape(B, P, V, 4).permute(0, 3, 1, 2).contiguous()                      # the reference's (B, C, P, V) input
    ref = sit_oracle.SiT(**kw)
    ref.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})

    def part(mod, lo, hi):
        return ((mod(xp[lo:hi]).squeeze(-1) - y[lo:hi]) ** 2).sum() / B

    l_ref, g_ref = _oracle_grads_cpu_chunked("cfg3", ref, B, 8, part)
    # engine.make_engine: what bench.py builds -- for this width TWO concurrent half-batch steps on two streams (round 6)
    eng = engine.make_engine(m, B, input_layout="surface", lr=0.0, momentum=0.0, keep_grads=True)
    assert isinstance(eng, engine.SplitTrainEngine) and eng.use_graph, "not the benchmarked launch form of this width"
    eng.step(x.to(DEV), y.to(DEV))                       # eager run + capture
    loss = float(eng.step())                             # the replayed graphs, same batch (lr = 0: same gradients)
    case = "engine/cfg3_b32_d12_oracle"
    check(case, "loss", dtype, abs(loss - l_ref) / abs(l_ref), "loss")
    worst_n, worst_e = _worst_grads(m.named_parameters(), g_ref)
    print("worst gradient norm:", worst_n, " worst gradient (element-wise, relative to the tensor):", worst_e)
    check(case, "gnorm", dtype, worst_n[0], "grad")
    check(case, "grad_rel", dtype, worst_e[0], "grad")
    assert eng.nonfinite_count == 0


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_mpp_engine_config5_benchmarked_form_against_cpu_oracle(pk, dtype):
    """the engine form `bench.py --model base --patches 1280 --batch 32 --task mpp` times (`also.cfg5`,
    BASELINE config 5's per-GPU share) -- SiT-base masked patch pre-training, 1280 patches x 45 vertices, DEPTH 12, raw
    surfaces, the default launch form (one hipGraph per step, device-side Philox draws, fused gather + corruption) -- one
    REPLAYED step with kept gradients against the CPU oracle replaying THAT step's draws (models/mpp.py:77-134 restated in
    oracle/sit_oracle.py; the loop of tools/pretrain.py:309-319).  B = 32, the benchmarked per-GPU batch: 26.8 TFLOP on the
    CPU in chunks of 4 samples -- about a minute on the box's 16 cores, which is the largest this test affords (the oracle
    run is shared by the two 16-bit modes: same seed, same Philox stream, asserted).  f16: fixed 1e-3 on the loss, every
    gradient's norm and every gradient element-wise; bf16: recorded bars under the fixed ceilings."""
    sit, mpp, engine = pk
    B, P, V = 32, 1280, 45
    kw = dict(sit_oracle.MODEL_SIZES["base"], num_patches=P, num_vertices=V, num_channels=4)
    assert kw["depth"] == 12
    torch.manual_seed(77)                                # seeds the engine's Philox stream (torch.initial_seed())
    model = sit.SiT(**kw, compute_dtype=dtype)
    model.allow_synthetic_table = True
    ssl = mpp.masked_patch_pretraining(model, 768, 4 * V, "cpu", mask_prob=0.75, replace_prob=0.8, swap_prob=0.02,
                                       channels=4, num_vertices=V)
    _load(ssl, 9)
    g = torch.Generator().manual_seed(3)
    x = torch.randn((B, 40962, 4), generator=g)
    eng = engine.TrainEngine(ssl, B, task="mpp", input_layout="surface", lr=0.0, momentum=0.0, keep_grads=True)
    assert eng.use_graph and not eng._overlap, "not the benchmarked launch form of this width"
    eng.step(x.to(DEV))                                  # eager run + capture (first draws)
    loss = float(eng.step())                             # the replayed graph (second draws)
    rnd = {k: v.cpu() for k, v in eng.last_randoms.items()}
    assert int(rnd["corrupted_sequence"].sum()) == B * 960
    from sitk import tables
    table = tables.load_table(P, V, allow_synthetic=True)
    tok = torch.from_numpy(sit_oracle.gather_tokens(x.numpy(), table))
    xp = tok.reshape(B, P, V, 4).permute(0, 3, 1, 2).contiguous()
    ref = sit_oracle.MaskedPatchPretraining(sit_oracle.SiT(**kw), 768, 4 * V, "cpu", mask_prob=0.75, replace_prob=0.8,
                                            swap_prob=0.02, channels=4, num_vertices=V)
    ref.load_state_dict({k: v.detach().clone() for k, v in ssl.state_dict().items()})

    def part(mod, lo, hi):                               # every sample masks exactly ceil(0.75 P) patches: equal denominators
        return mod(xp[lo:hi], randoms={k: v[lo:hi] for k, v in rnd.items()})[0] * ((hi - lo) / B)

    import hashlib
    key = "cfg5-" + hashlib.sha256(b"".join(rnd[k].numpy().tobytes() for k in sorted(rnd))).hexdigest()
    l_ref, g_ref = _oracle_grads_cpu_chunked(key, ref, B, 4, part)
    case = "engine/cfg5_mpp_b32_d12_oracle"
    check(case, "loss", dtype, abs(loss - l_ref) / abs(l_ref), "loss")
    worst_n, worst_e = _worst_grads(ssl.named_parameters(), g_ref)
    print("worst gradient norm:", worst_n, " worst gradient (element-wise, relative to the tensor):", worst_e)
    check(case, "gnorm", dtype, worst_n[0], "grad")
    check(case, "grad_rel", dtype, worst_e[0], "grad")
    assert eng.nonfinite_count == 0


@pytest.mark.parametrize("size,dtype,optimizer", [("small", "f32", "sgd"), ("small", "bf16", "sgd"), ("small", "f16", "sgd"),
                                                  ("small", "bf16", "adamw"), ("tiny", "bf16", "sgd"), ("tiny", "f16", "sgd")])
def test_split_engine_matches_whole_batch_engine(pk, size, dtype, optimizer):
    """Round 6: engine.SplitTrainEngine -- the batch as two concurrent half-batch steps on two streams, two gradient buffers, ONE
    optimizer pass over both (sitk_*_step_dev with grad2; the loss and every gradient are the means of the halves') -- against the
    whole-batch TrainEngine on the same batches: three steps of SGD(momentum 0.9, weight decay) / AdamW from the same weights
    (tools/train.py:280-291).  Every per-sample quantity is computed by the same kernels on the same operands; what differs is the
    ORDER of the fp32 sums over samples (weight gradients, LayerNorm parameter gradients, column sums) and, in f16, the loss
    scale each half picks: losses to 2e-6, every parameter's three-step update to 2e-4 of its norm."""
    sit, _, engine = pk
    B, lr = 8, 1e-5                                       # (bench.py's rate: a stable trajectory, so that the forms can be compared)
    kw = dict(sit_oracle.MODEL_SIZES[size], num_patches=320, num_vertices=153, num_channels=4)   # (tiny: the fused dim-192 kernels)
    kw["depth"] = 2
    g = torch.Generator(device=DEV).manual_seed(11)
    xs = [torch.randn((B, 40962, 4), device=DEV, generator=g) for _ in range(3)]
    ys = [torch.randn((B,), device=DEV, generator=g) * 2 + 40 for _ in range(3)]
    res = {}
    for form in ("whole", "split"):
        m = sit.SiT(**kw, compute_dtype=dtype)
        _load(m, 33)
        before = {k: p.detach().clone() for k, p in m.named_parameters()}
        okw = dict(input_layout="surface", optimizer=optimizer, lr=lr, momentum=0.9, weight_decay=1e-2)
        eng = engine.TrainEngine(m, B, use_graph=True, **okw) if form == "whole" else engine.SplitTrainEngine(m, B, **okw)
        losses = [float(eng.step(x, y)) for x, y in zip(xs, ys)]
        torch.cuda.synchronize()
        res[form] = (losses, {k: p.detach().cpu() - before[k].cpu() for k, p in m.named_parameters()})
        assert eng.nonfinite_count == 0 and eng.fp.still_flat()
        if form == "split":
            assert all(pr[-1]["ok"] for pr in eng.stream_probe), eng.stream_probe
            # consumed gradients are cleared in BOTH buffers (the loss and the other accumulators behind them too)
            assert float(eng.owner.fp.grad_all.abs().max()) == 0.0 and float(eng.other.fp.grad_all.abs().max()) == 0.0
    (lw, uw), (ls, us) = res["whole"], res["split"]
    assert max(abs(a - b) / abs(b) for a, b in zip(ls, lw)) < 2e-6, (ls, lw)
    worst = max((rel(us[k], uw[k]), k) for k in uw)
    print("worst three-step update, split vs whole:", worst)
    assert worst[0] < 2e-4, worst


def test_split_engine_kept_gradients_and_reproducibility(pk):
    """SplitTrainEngine(keep_grads=True): after step() the parameters' .grad hold the gradient of the WHOLE batch's loss (the mean
    of the halves'), as the whole-batch engine leaves it; and two split engines from the same weights agree from run to run
    although their halves' streams interleave differently every time."""
    sit, _, engine = pk
    B = 8
    kw = dict(sit_oracle.MODEL_SIZES["small"], num_patches=320, num_vertices=153, num_channels=4)
    kw["depth"] = 2
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((B, 40962, 4), device=DEV, generator=g)
    y = torch.randn((B,), device=DEV, generator=g) * 2 + 40
    grads = {}
    for form in ("whole", "split"):
        m = sit.SiT(**kw, compute_dtype="bf16")
        _load(m, 33)
        eng = (engine.TrainEngine(m, B, input_layout="surface", lr=0.0, momentum=0.0, keep_grads=True, use_graph=True) if form == "whole"
               else engine.SplitTrainEngine(m, B, input_layout="surface", lr=0.0, momentum=0.0, keep_grads=True))
        eng.step(x, y)
        loss = float(eng.step())
        grads[form] = (loss, {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()})
    assert abs(grads["split"][0] - grads["whole"][0]) / abs(grads["whole"][0]) < 2e-6
    worst = max((rel(grads["split"][1][k], grads["whole"][1][k]), k) for k in grads["whole"][1])
    print("worst kept gradient, split vs whole:", worst)
    assert worst[0] < 1e-5, worst
    # run to run: the halves' streams interleave differently every time, the results must not depend on it.  (Bit equality is not
    # asked for: at this width the step has order-dependent last bits of its own -- DESIGN.md section 2 lists them -- with or without
    # the split; what the split adds, the sum of the two gradient buffers, has a fixed order.)
    runs = []
    for _ in range(2):
        m = sit.SiT(**kw, compute_dtype="bf16")
        _load(m, 33)
        before = {k: p.detach().clone() for k, p in m.named_parameters()}
        eng = engine.SplitTrainEngine(m, B, input_layout="surface", lr=1e-5, momentum=0.9)
        losses = [float(eng.step(x, y)) for _ in range(3)]
        torch.cuda.synchronize()
        runs.append((losses, {k: p.detach().cpu() - before[k].cpu() for k, p in m.named_parameters()}))
    assert max(abs(a - b) / abs(b) for a, b in zip(runs[0][0], runs[1][0])) < 1e-6, (runs[0][0], runs[1][0])
    worst = max((rel(runs[0][1][k], runs[1][1][k]), k) for k in runs[0][1])
    print("worst three-step update, run to run:", worst)
    assert worst[0] < 1e-5, worst


def test_engine_config3_width_bf16_against_f32_mode(pk):
    """BASELINE config 3 at its full width -- SiT-small, 1280 patches (N = 1281), B = 32 -- through the long-sequence ring
    attention and the two-per-CU N % 192 GEMMs, depth 2: the bf16 engine step against the SAME engine in f32 compute
    mode (which the goldens pin to 1e-6 at B = 1; models/sit.py:57-80, tools/train.py:280-291).  Loss and the parameter
    update of every tensor after one step."""
    sit, _, engine = pk
    B, lr = 32, 0.01
    kw = dict(sit_oracle.MODEL_SIZES["small"], num_patches=1280, num_vertices=45, num_channels=4)
    kw["depth"] = 2
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn((B, 4, 1280, 45), device=DEV, generator=g)
    y = torch.randn((B,), device=DEV, generator=g) * 2 + 40
    res = {}
    for dtype in ("f32", "bf16"):
        m = sit.SiT(**kw, compute_dtype=dtype)
        _load(m, 33)
        before = {k: p.detach().clone() for k, p in m.named_parameters()}
        eng = engine.TrainEngine(m, B, input_layout="patched", lr=lr, momentum=0.9, use_graph=False)
        loss = float(eng.step(x, y))
        res[dtype] = (loss, {k: p.detach().cpu() - before[k] for k, p in m.named_parameters()})
    l32, u32 = res["f32"]
    l16, u16 = res["bf16"]
    check("engine/cfg3_b32_d2", "loss", "bf16", abs(l16 - l32) / abs(l32), "out")
    worst = max((rel(u16[k], u32[k]), k) for k in u32)
    print("worst parameter update:", worst)
    check("engine/cfg3_b32_d2", "update_rel", "bf16", worst[0], "grad")


@pytest.mark.parametrize("h16", ["bf16", "f16"])
def test_mpp_engine_config5_width_16bit_against_f32_mode(pk, h16):
    """BASELINE config 5's per-GPU share at its full width -- SiT-base MPP, 1280 patches x 45 vertices, 32 samples --
    depth 1: the 16-bit MPP engine step (device draws, fused gather + corruption, padded to_original, masked loss, weight
    gradients in the batched launch, mask_token gradient) against the same engine in f32 compute mode on the SAME draws
    (models/mpp.py:77-134).  Loss and every gradient.  f16: the fixed 1e-3 bars, nothing may overflow behind the loss scale
    (d mask_token sums ~25 k loss-scaled rows: its 1 x D x K product runs in the f32 kernel there) and the optimizer's
    non-finite counter stays 0."""
    sit, mpp, engine = pk
    B, P, V = 32, 1280, 45
    kw = dict(sit_oracle.MODEL_SIZES["base"], depth=1, num_patches=P, num_vertices=V, num_channels=4)
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn((B, 4, P, V), device=DEV, generator=g)
    res = {}
    for dtype in ("f32", h16):
        model = sit.SiT(**kw, compute_dtype=dtype)
        model.allow_synthetic_table = True
        ssl = mpp.masked_patch_pretraining(model, 768, 4 * V, "cpu", mask_prob=0.75, replace_prob=0.8, swap_prob=0.02,
                                           channels=4, num_vertices=V)
        _load(ssl, 9)
        eng = engine.TrainEngine(ssl, B, task="mpp", input_layout="patched", lr=0.0, momentum=0.0, use_graph=False,
                                 keep_grads=True)
        loss = float(eng.step(x))
        rnd = eng.last_randoms
        res[dtype] = (loss, {k: p.grad.detach().cpu().clone() for k, p in ssl.named_parameters()}, rnd)
        assert eng.nonfinite_count == 0 and all(bool(torch.isfinite(v).all()) for v in res[dtype][1].values())
    l32, g32, r32 = res["f32"]
    l16, g16, r16 = res[h16]
    for k in r32:
        assert torch.equal(r32[k], r16[k]), k                  # same Philox stream in both modes
    assert int(r32["corrupted_sequence"].sum()) == B * 960
    check("engine/cfg5_mpp_b32_d1", "loss", h16, abs(l16 - l32) / abs(l32), "out")
    worst = max((rel(g16[k], g32[k]), k) for k in g32 if float(g32[k].abs().max()) > 0)
    print("worst gradient:", worst)
    check("engine/cfg5_mpp_b32_d1", "grad_rel", h16, worst[0], "grad")


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_optimizer_skips_and_counts_nonfinite_gradients(pk, optimizer):
    """A gradient element that is not finite (an f16 intermediate that overflowed behind the loss scale) must not reach the
    parameters or the optimizer state: the fused optimizer pass skips it, zeroes it like every consumed gradient and counts
    it; every other element is updated as usual (tools/train.py:291's optimizer.step() has no such guard: the reference
    computes in fp32)."""
    from sitk import runtime as rt
    n = 4096 + 64
    g0 = torch.Generator(device=DEV).manual_seed(5)
    p = torch.randn(n, device=DEV, generator=g0)
    g = torch.randn(n, device=DEV, generator=g0)
    g[5], g[1000], g[4100] = float("inf"), float("nan"), float("-inf")
    p0, gref = p.clone(), g.clone()
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    hyper = torch.tensor([0.1, 1.0, 1.0, 0.0], dtype=torch.float64, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    if optimizer == "sgd":
        rt.check(rt.lib.sitk_sgd_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), n, hyper.data_ptr(), 0.9, 0.0, 0, 1.0, 1, 0, -1,
                                          None, None, cnt.data_ptr(), None, None, 1.0, rt.stream_ptr()))
        bad = ~torch.isfinite(gref)                              # per ELEMENT (ABI 10; ABI 9 skipped the 16-byte vector)
        ref = p0 - 0.1 * gref
        assert int(cnt) == 3
    else:
        rt.check(rt.lib.sitk_adam_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, hyper.data_ptr(), 0.9, 0.999,
                                           1e-8, 0.0, 0, 1.0, 1, 0, -1, None, None, cnt.data_ptr(), None, None, 1.0, rt.stream_ptr()))
        bad = ~torch.isfinite(gref)
        ref = p0 - 0.1 * torch.sign(gref)                        # first Adam step: m / sqrt(v) = sign(g)
        assert int(cnt) == 3
    assert torch.equal(p[bad], p0[bad]) and float(m[bad].abs().max()) == 0.0
    assert bool(torch.isfinite(p).all()) and bool(torch.isfinite(m).all())
    assert float((p[~bad] - ref[~bad]).abs().max()) < 1e-4
    assert float(g.abs().max()) == 0.0                            # consumed gradients are zeroed, the skipped ones too
    # nonfinite = NULL (what the engine passes in bf16 / f32): no guard, like optimizer.step() of tools/train.py:291 -- the
    # non-finite elements reach their parameters (and only theirs)
    p, g, m, v = p0.clone(), gref.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    hyper = torch.tensor([0.1, 1.0, 1.0, 0.0], dtype=torch.float64, device=DEV)
    if optimizer == "sgd":
        rt.check(rt.lib.sitk_sgd_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), n, hyper.data_ptr(), 0.9, 0.0, 0, 1.0, 1, 0, -1,
                                          None, None, None, None, None, 1.0, rt.stream_ptr()))
    else:
        rt.check(rt.lib.sitk_adam_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, hyper.data_ptr(), 0.9, 0.999,
                                           1e-8, 0.0, 0, 1.0, 1, 0, -1, None, None, None, None, None, 1.0, rt.stream_ptr()))
    assert not bool(torch.isfinite(p[bad]).any()) and bool(torch.isfinite(p[~bad]).all())
    assert float((p[~bad] - ref[~bad]).abs().max()) < 1e-4


@pytest.mark.parametrize("optimizer", ["sgd", "adam", "adamw"])
def test_graph_follows_lr_schedule_and_step_count(pk, optimizer):
    """tools/pretrain.py:42-50 change the learning rate between steps and tools/train.py:228-241 use Adam / AdamW: the
    captured step must follow both (lr and Adam's bias corrections are read from device memory).  Three steps with
    the learning rate changed before the second one: hipGraph replay == eager engine == torch.optim on the autograd path."""
    sit, _, engine = pk
    B = 2
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], depth=2, num_patches=80, num_vertices=561, num_channels=4)
    base = sit.SiT(**kw, compute_dtype="f32")
    _load(base, 17)
    x = torch.from_numpy(detgen.normal("lr/x", (B, 4, 80, 561), seed=1)).to(DEV)
    y = torch.from_numpy(detgen.normal("lr/y", (B,), seed=1)).to(DEV)
    lrs = [1e-3, 4e-3, 4e-3]
    okw = dict(weight_decay=0.01) if optimizer != "sgd" else dict(momentum=0.9)
    ref = copy.deepcopy(base).to(DEV)
    topt = {"sgd": torch.optim.SGD, "adam": torch.optim.Adam, "adamw": torch.optim.AdamW}[optimizer](ref.parameters(), lr=lrs[0], **okw)
    for lr in lrs:
        for gr in topt.param_groups:
            gr["lr"] = lr
        topt.zero_grad()
        torch.nn.functional.mse_loss(ref(x).squeeze(), y).backward()
        topt.step()
    flats = {}
    for use_graph in (False, True):
        m = copy.deepcopy(base)
        eng = engine.TrainEngine(m, B, input_layout="patched", optimizer=optimizer, lr=lrs[0], use_graph=use_graph, **okw)
        for lr in lrs:
            eng.set_lr(lr)
            eng.step(x, y)
        if use_graph:
            assert eng._graphs and "step" in eng._graphs, "the optimizer must run from the captured graph"
        flats[use_graph] = eng.fp.flat.clone()
        worst = max((rel(p.data, q.data), k) for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()))
        assert worst[0] < 2e-5, worst
    assert rel(flats[True], flats[False]) < 1e-6      # (f32 verification mode: its generic 64 x 64 weight-gradient tiles add with float atomics)


def test_resident_dataset_pipeline(pk):
    """SURVEY 8(f).2 (tools/train.py:97-113,282; tools/preprocessing.py:72): raw surfaces resident in HBM, per-channel
    normalisation + sample selection inside the gather, labels gathered alongside -- equal to an engine fed the
    numpy-normalised batch directly (the gather itself is bit exact: test_gather_with_fused_normalisation_bit_exact); only the int32 indices are copied per step."""
    sit, _, engine = pk
    S, B = 7, 3
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], depth=1, num_patches=320, num_vertices=153, num_channels=4)
    base = sit.SiT(**kw, compute_dtype="f32")
    _load(base, 9)
    raw = detgen.normal("ds/x", (S, 40962, 4), mean=3.0, std=5.0, seed=1)
    labels = detgen.normal("ds/y", (S, 1), mean=40.0, std=2.0, seed=1)
    mean, std = raw.reshape(-1, 4).mean(0).astype(np.float32), raw.reshape(-1, 4).std(0).astype(np.float32)
    normed = ((raw - mean) / std).astype(np.float32)                   # tools/preprocessing.py:72 in fp32
    picks = [np.array([5, 0, 3]), np.array([6, 6, 1])]
    for use_graph in (False, True):
        e1 = engine.TrainEngine(copy.deepcopy(base), B, input_layout="surface", lr=0.01, use_graph=use_graph)
        e2 = engine.TrainEngine(copy.deepcopy(base), B, input_layout="surface", lr=0.01, use_graph=use_graph,
                                normalise=(mean, std))
        e2.load_dataset(raw, labels)
        for idx in picks:
            l1 = e1.step(torch.from_numpy(normed[idx]).to(DEV), torch.from_numpy(labels[idx]).to(DEV)).clone()
            l2 = e2.step(indices=idx).clone()
            assert abs(float(l1) - float(l2)) <= 1e-6 * abs(float(l1)), (float(l1), float(l2))
        assert rel(e2.fp.flat, e1.fp.flat) < 1e-6        # same tokens bit for bit; float atomics in the loss / column sums
        assert torch.equal(e2.target.cpu(), torch.from_numpy(labels[picks[-1]]))


@pytest.mark.parametrize("feed", ["batches", "indices_host", "indices_device"])
def test_prefetched_gather_sees_every_new_batch(pk, feed):
    """With a side stream the patch gather of step t + 1 is enqueued there and runs beside step t's tail (two token buffers;
    load_batch() / the index copy order it behind the new input).  A DIFFERENT batch every step, fed through step(x, y) or
    through a resident data set with host / device indices: losses and parameters bit-equal to an engine whose gather sits
    in front of the patch embedding on the main stream."""
    sit, _, engine = pk
    S, B, steps = 12, 8, 5
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], depth=3, num_patches=320, num_vertices=153, num_channels=4)
    base = sit.SiT(**kw, compute_dtype="bf16")
    _load(base, 17)
    g = torch.Generator(device=DEV).manual_seed(11)
    xs = torch.randn((S, 40962, 4), device=DEV, generator=g)
    ys = torch.randn((S, 1), device=DEV, generator=g) + 40
    picks = [torch.randperm(S, generator=torch.Generator().manual_seed(i))[:B] for i in range(steps)]
    out = []
    for prefetch in (True, False):
        eng = engine.TrainEngine(copy.deepcopy(base), B, input_layout="surface", lr=1e-3, momentum=0.9, prefetch_gather=prefetch)
        assert bool(eng._overlap) and eng._prefetch == prefetch
        if feed != "batches":
            eng.load_dataset(xs, ys)
        losses = []
        for idx in picks:
            if feed == "batches":
                losses.append(eng.step(xs[idx.to(DEV)], ys[idx.to(DEV)]).clone())
            else:
                losses.append(eng.step(indices=idx.to(DEV) if feed == "indices_device" else idx.numpy()).clone())
        torch.cuda.synchronize()
        out.append((torch.cat(losses), eng.fp.flat.clone()))
    assert torch.equal(out[0][0], out[1][0]), (out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])
    assert len(set(float(v) for v in out[0][0])) == steps          # the batches did differ


@pytest.mark.parametrize("tscale", [1e-4, 1.0, 1e4])
def test_f16_loss_scale_follows_the_batch(pk, tscale):
    """f16 compute mode: the gradient stream is scaled by a power of two the fused head + loss call picks from THIS batch's
    largest |d loss / d logits| (tests of the mode at ordinary magnitudes are above).  Targets 1e-4 .. 1e4 of the ordinary
    size move that gradient over eight decades -- far outside what a fixed scale and IEEE half's 5 exponent bits could
    carry -- and the gradients must still match the f32-mode engine; the published scale puts the batch maximum in [64, 128)."""
    sit, _, engine = pk
    B = 4
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], depth=2, num_patches=320, num_vertices=153, num_channels=4)
    base = sit.SiT(**kw, compute_dtype="f32")
    _load(base, 11)
    x = torch.from_numpy(detgen.normal("ls/x", (B, 4, 320, 153), seed=1)).to(DEV)
    # (targets of ONE sign: with mixed signs the samples' loss gradients cancel in the batch sum -- y = -2, -5, -10, +15 leaves a
    # twentieth of the terms -- and every mode's relative gradient error grows by that factor, bf16's and f16's alike)
    y = -(1.0 + torch.from_numpy(detgen.normal("ls/y", (B,), seed=1)).to(DEV).abs()) * tscale
    grads = {}
    for dtype in ("f32", "f16"):
        m = sit.SiT(**kw, compute_dtype=dtype)
        m.load_state_dict(base.state_dict())
        eng = engine.TrainEngine(m, B, input_layout="patched", lr=0.0, momentum=0.0, use_graph=False, keep_grads=True)
        loss = float(eng.step(x, y))
        assert np.isfinite(loss)
        grads[dtype] = (loss, {k: p.grad.detach().clone() for k, p in m.named_parameters()})
        if dtype == "f16":
            S = float(eng.gscale[0])
            dl_max = float((2.0 * (eng.logits.view(-1) - y) / B).abs().max())
            assert S == 2.0 ** round(np.log2(S)) and 64.0 <= dl_max * S < 128.0, (S, dl_max)
    assert abs(grads["f16"][0] - grads["f32"][0]) <= 1e-3 * abs(grads["f32"][0])
    worst = max((rel(grads["f16"][1][k], g), k) for k, g in grads["f32"][1].items())
    print("worst gradient:", worst)
    assert all(bool(torch.isfinite(g).all()) for g in grads["f16"][1].values())
    assert worst[0] < 2e-3, worst
