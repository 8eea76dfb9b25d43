"""CPU (gloo, world_size 2) tests of the data-parallel logic used by sitk.engine.TrainEngine:
the flat-buffer bucket ranges cover every gradient exactly once in backward order, and
"all-reduce(sum) of shard gradients, scaled by 1/world" equals the full-batch gradient for the
batch-mean losses of the path (tools/train.py:246, models/mpp.py:132).  The HIP kernels are not
involved (no GPU here): the gradients come from the CPU oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import detgen, sit_oracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cpu_flat_params(module):
    """sitk.engine.FlatParams on the CPU (it only needs torch): same offsets as on the GPU."""
    import sitk  # noqa: F401
    from sitk import engine
    return engine.FlatParams(module, "cpu")


def _mpp_module(depth):
    import sitk  # noqa: F401
    from sitk.models import mpp, sit
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], depth=depth, num_patches=80, num_vertices=561, num_channels=4)
    model = sit.SiT(**kw)
    return mpp.masked_patch_pretraining(model, 192, 4 * 561, "cpu", mask_prob=0.75, replace_prob=0.8, swap_prob=0.02,
                                        channels=4, num_vertices=561)


def _writer_stage(name, slices, task, head_deferred=False):
    """The spec, restated from the step's launch order (not from sitk.engine's code): index of the segment whose
    kernels write this parameter's gradient LAST; len(slices) = the finish stage."""
    finish = len(slices)
    name = name.removeprefix("transformer.") if task == "mpp" else name
    if name.startswith("transformer.layers."):
        layer = int(name.split(".")[2])
        return next(i for i, (lb, le) in enumerate(slices) if lb <= layer < le)
    if name.startswith("mlp_head."):
        # fused head + loss kernel, segment 0 (never written under MPP: zero); with a side stream the sum of its per-sample
        # terms runs in the finish stage (sitk_head_loss_fwd_bwd_deferred + sitk_head_finalize)
        return finish if head_deferred else 0
    # to_original.* (weight-gradient launch of the slice ending at layer 0, or finish), mask_token, cls_token,
    # pos_embedding, to_patch_embedding.1.*: all behind the last backward slice
    return finish


@pytest.mark.parametrize("head_deferred", [False, True])
@pytest.mark.parametrize("task", ["regression", "mpp"])
@pytest.mark.parametrize("nsl", [1, 2, 3, 4, 12])
def test_bucket_plan_reduces_every_gradient_once_and_only_after_it_is_written(task, nsl, head_deferred):
    """`to_original.*` sits BEHIND mlp_head in the flat buffer, so an offset-derived "everything from
    the first finished layer to the end of the buffer" range all-reduced it right after slice 0 -- two slices before
    the kernel that writes it.  The plan is derived from where each gradient is written instead; this walks
    named_parameters() of the real modules and checks (a) every float of the flat buffer is reduced exactly once,
    (b) no parameter is reduced at a point of the step that precedes the segment writing it."""
    import sitk  # noqa: F401
    from sitk import engine
    depth = 12
    ssl = _mpp_module(depth)
    module = ssl if task == "mpp" else ssl.transformer
    fp = _cpu_flat_params(module)
    bounds = [round(i * depth / nsl) for i in range(nsl + 1)]
    slices = [(bounds[i], bounds[i + 1]) for i in range(nsl)][::-1]
    plan = engine.grad_bucket_plan(fp, engine.grad_write_stages(module, task, slices, head_deferred=head_deferred), nsl)
    assert len(plan) == nsl
    flat_ranges = sorted(r for point in plan for r in point)
    assert flat_ranges[0][0] == 0 and flat_ranges[-1][1] == fp.total
    assert all(a[1] == b[0] for a, b in zip(flat_ranges, flat_ranges[1:])), flat_ranges          # (a)
    for name, p in module.named_parameters():
        lo, n = fp.offsets[id(p)]
        point = next(i for i, rs in enumerate(plan) if any(a <= lo and lo + n <= b for a, b in rs))
        # point i < nsl - 1 is issued right after segment i; point nsl - 1 after the finish stage
        issued_after = point if point < nsl - 1 else nsl
        assert issued_after >= _writer_stage(name, slices, task, head_deferred), (name, point, slices)           # (b)
    if task == "mpp" and nsl > 1:
        lo, _ = fp.offsets[id(ssl.to_original.weight)]
        assert not any(a <= lo < b for a, b in plan[0]), "to_original reduced with slice 0 again"


def _side_plan_cases():
    """(task, depth, side layers, optimizer scope, side launches per bucket): every combination that exists -- optimize='sit' is an
    MPP option, explicit bucket sizes must not cover more side launches than the depth makes."""
    cases = []
    for depth, side in [(12, 8), (4, 3), (2, 1), (6, 4)]:
        launches = (side + 1) // 2
        for per_bucket in [1, 2, 3, [1], [2, 1], [launches - 1, 1] if launches >= 2 else [1]]:
            if not isinstance(per_bucket, int) and sum(per_bucket) > launches:
                continue
            for task, optimize in [("regression", "all"), ("mpp", "all"), ("mpp", "sit")]:
                cases.append((task, depth, side, optimize, per_bucket))
    return cases


@pytest.mark.parametrize("task,depth,side,optimize,per_bucket", _side_plan_cases())
def test_side_launch_bucket_plan_is_one_range_per_launch_and_never_early(task, depth, side, optimize, per_bucket):
    """Round 5: the data-parallel form of the fused path all-reduces one bucket per SIDE LAUNCH of
    sitk_encoder_bwd_overlap.  The spec, restated from include/sitk.h (ABI 10) and csrc/encoder.hip's launch order, not from
    sitk.engine: side launch i carries the Linear weight + bias gradients of layers depth - 1 - 2 i and depth - 2 - 2 i (a
    single layer last when `side` is odd); every LayerNorm parameter, the tail launch's layers, the patch embedding,
    cls_token, pos_embedding, mlp_head.*, to_original.*, mask_token are final only behind the finish stage.  Checks: (a) the
    flat buffer ordered by write stage makes every bucket ONE contiguous range; (b) ranges are disjoint and cover [0, n_opt);
    (c) no parameter sits in a bucket issued before its writer; (d) parameters outside the optimizer's scope are not reduced."""
    import sitk  # noqa: F401
    from sitk import engine
    ssl = _mpp_module(depth)
    module = ssl if task == "mpp" else ssl.transformer
    sit = ssl.transformer
    groups = engine.side_launch_groups(0, depth, side)
    want_groups, top = [], depth
    while depth - top < side:
        n = min(2, side - (depth - top))
        want_groups.append(list(range(top - n, top)))
        top -= n
    assert groups == want_groups and sum(len(g) for g in groups) == side
    stage = engine.grad_write_stages_side(module, task, groups, per_bucket)
    sizes = engine.side_bucket_sizes(len(groups), per_bucket)
    n_early = len(sizes)
    first = [sum(sizes[:b]) for b in range(n_early)]            # first launch of every early bucket
    frozen = set()
    if task == "mpp":
        frozen |= {id(p) for p in sit.mlp_head.parameters()}
        if optimize == "sit":
            frozen |= {id(p) for p in ssl.to_original.parameters()} | {id(ssl.mask_token)}
    fp = engine.FlatParams(module, "cpu", order=lambda p: (id(p) in frozen, stage[id(p)]))
    n_opt = min([fp.offsets[i][0] for i in frozen], default=fp.total)
    plan = engine.grad_bucket_plan(fp, stage, n_early + 1, limit=n_opt)
    assert len(plan) == n_early + 1
    assert all(len(rs) == 1 for rs in plan), plan                                                   # (a)
    flat_ranges = sorted(r for point in plan for r in point)
    assert flat_ranges[0][0] == 0 and flat_ranges[-1][1] == n_opt
    assert all(a[1] == b[0] for a, b in zip(flat_ranges, flat_ranges[1:])), flat_ranges              # (b)
    prefix = "transformer." if task == "mpp" else ""
    for name, p in module.named_parameters():
        lo, n = fp.offsets[id(p)]
        if id(p) in frozen:
            assert lo >= n_opt, name                                                                # (d)
            continue
        point = next(i for i, rs in enumerate(plan) if any(a <= lo and lo + n <= b for a, b in rs))
        short = name.removeprefix(prefix)
        writer = n_early                                         # finish stage
        if short.startswith("transformer.layers.") and ".norm." not in short:
            layer = int(short.split(".")[2])
            launch = next((i for i, g in enumerate(groups) if layer in g), None)
            # bucket b covers launches first[b] .. first[b] + sizes[b] - 1 and is issued behind the last of them; a launch no
            # early bucket covers travels with the final bucket (behind the finish stage, which joins the side stream)
            writer = n_early
            if launch is not None:
                writer = next((b for b in range(n_early) if first[b] <= launch < first[b] + sizes[b]), n_early)
        assert point == writer, (name, point, writer)                                               # (c) (and not late either)
    if task == "mpp" and optimize == "all":
        assert fp.offsets[id(ssl.to_original.weight)][0] < n_opt


def test_flat_params_order_keeps_module_views_and_values():
    """FlatParams(order=...) only permutes the flat storage: every parameter keeps its values and is a view of its own range."""
    import sitk  # noqa: F401
    from sitk import engine
    ssl = _mpp_module(2)
    before = {n: p.detach().clone() for n, p in ssl.named_parameters()}
    names = {id(p): n for n, p in ssl.named_parameters()}
    fp = engine.FlatParams(ssl, "cpu", order=lambda p: -len(names[id(p)]))
    assert fp.still_flat()
    seen = torch.zeros(fp.total, dtype=torch.bool)
    for n, p in ssl.named_parameters():
        assert torch.equal(p.detach(), before[n]), n
        o, k = fp.offsets[id(p)]
        assert not bool(seen[o:o + k].any())
        seen[o:o + k] = True
        assert p.grad.data_ptr() == fp.grad.data_ptr() + 4 * o


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], depth=2, num_patches=80, num_vertices=561, num_channels=4)
    model = sit_oracle.SiT(**kw)
    vals = detgen.fill_state_dict(model.state_dict(), seed=3)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    B = 4
    x = torch.from_numpy(detgen.normal("dp/x", (B, 4, 80, 561), seed=1))
    y = torch.from_numpy(detgen.normal("dp/y", (B,), seed=1))
    shard = slice(rank * B // world, (rank + 1) * B // world)
    loss = torch.nn.functional.mse_loss(model(x[shard]).squeeze(-1), y[shard])
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    # bucketed all-reduce in backward order (3 ranges), then the optimizer's 1/world scale
    n = flat.numel()
    works = [dist.all_reduce(flat[lo:hi], async_op=True) for lo, hi in ((2 * n // 3, n), (n // 3, 2 * n // 3), (0, n // 3))]
    for w in works:
        w.wait()
    flat /= world
    if rank == 0:
        model.zero_grad()
        full = torch.nn.functional.mse_loss(model(x).squeeze(-1), y)
        full.backward()
        ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        q.put(float((flat - ref).norm() / ref.norm()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gradients_average_to_full_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-5, err


def test_side_bucket_sizes_and_launch_groups():
    """The two small rules the data-parallel engine builds its plan from (include/sitk.h, ABI 10): side launch i carries the
    `per_launch` layers below layer_end - per_launch i; early buckets are consecutive side launches, what a size list leaves
    uncovered travels with the final bucket."""
    import sitk  # noqa: F401
    from sitk import engine
    assert engine.side_launch_groups(0, 12, 8) == [[10, 11], [8, 9], [6, 7], [4, 5]]
    assert engine.side_launch_groups(0, 12, 9, 3) == [[9, 10, 11], [6, 7, 8], [3, 4, 5]]
    assert engine.side_launch_groups(0, 4, 3) == [[2, 3], [1]]
    assert engine.side_launch_groups(2, 6, 8, 1) == [[5], [4], [3], [2]]
    assert engine.side_bucket_sizes(4, 1) == [1, 1, 1, 1] and engine.side_bucket_sizes(4, 3) == [3, 1]
    assert engine.side_bucket_sizes(4, 4) == [4] and engine.side_bucket_sizes(4, 9) == [4]
    assert engine.side_bucket_sizes(4, [3]) == [3] and engine.side_bucket_sizes(4, [3, 1]) == [3, 1]
    for bad in ([0], [3, 2], [5]):
        with pytest.raises(ValueError):
            engine.side_bucket_sizes(4, bad)


@pytest.mark.parametrize("nsl", [1, 3])
def test_slice_bucket_plan_leaves_the_frozen_parameters_out(nsl):
    """TrainEngine(task='mpp', optimize='sit') on the slice form (other widths than 192, use_graph=True): the parameters outside the
    optimizer's scope sit at the END of the flat buffers and no all-reduce range reaches them; everything else is reduced exactly
    once (tools/pretrain.py:267-280: they never receive an update, so their gradients need no reduction either)."""
    import sitk  # noqa: F401
    from sitk import engine
    depth = 6
    ssl = _mpp_module(depth)
    sit = ssl.transformer
    frozen = {id(p) for p in sit.mlp_head.parameters()} | {id(p) for p in ssl.to_original.parameters()} | {id(ssl.mask_token)}
    fp = engine.FlatParams(ssl, "cpu", order=lambda p: (id(p) in frozen, 0))
    n_opt = min(fp.offsets[i][0] for i in frozen)
    bounds = [round(i * depth / nsl) for i in range(nsl + 1)]
    slices = [(bounds[i], bounds[i + 1]) for i in range(nsl)][::-1]
    plan = engine.grad_bucket_plan(fp, engine.grad_write_stages(ssl, "mpp", slices), nsl, limit=n_opt)
    ranges = sorted(r for point in plan for r in point)
    assert ranges[0][0] == 0 and ranges[-1][1] == n_opt and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    for name, p in ssl.named_parameters():
        lo, n = fp.offsets[id(p)]
        inside = any(a <= lo and lo + n <= b for a, b in ranges)
        assert inside == (id(p) not in frozen), name
        assert (lo >= n_opt) == (id(p) in frozen), name
