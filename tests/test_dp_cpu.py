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


def test_bucket_ranges_cover_flat_buffer_once():
    """Same arithmetic as TrainEngine._grad_range_after / step(): slices run last layer first, the
    last range is widened down to offset 0 (embedding, cls, pos gradients finish last)."""
    depth, per_layer, head, embed = 12, 1000, 50, 300
    offsets = [embed + l * per_layer for l in range(depth)]
    total = embed + depth * per_layer + head
    for nsl in (1, 2, 3, 4, 12):
        bounds = [round(i * depth / nsl) for i in range(nsl + 1)]
        slices = [(bounds[i], bounds[i + 1]) for i in range(nsl)][::-1]
        ranges = []
        for i, (lb, _) in enumerate(slices):
            lo = offsets[lb]
            hi = total if i == 0 else offsets[slices[i - 1][0]]
            if i < len(slices) - 1:
                ranges.append((lo, hi))
            else:
                ranges.append((0, hi))
        covered = sorted(ranges)
        assert covered[0][0] == 0 and covered[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:])), (nsl, covered)


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
