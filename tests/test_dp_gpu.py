"""Data-parallel TrainEngine on real kernels: 2 ranks (both on cuda:0, gloo transport -- the box has a
single GPU and RCCL refuses two ranks on one device) must reproduce the single-process full-batch
update: the bucketed, slice-overlapped gradient all-reduce + 1/world scaling of sitk.engine is
backend-agnostic, so this exercises exactly the code path bench.py runs over RCCL."""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import detgen, sit_oracle  # noqa: E402

KW = dict(sit_oracle.MODEL_SIZES["tiny"], depth=4, num_patches=80, num_vertices=561, num_channels=4)
B, STEPS, LR = 4, 2, 0.05


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model(dtype):
    import sitk  # noqa: F401
    from sitk.models.sit import SiT
    m = SiT(**KW, compute_dtype=dtype)
    vals = detgen.fill_state_dict(m.state_dict(), seed=13)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    return m


def _data():
    x = torch.from_numpy(detgen.normal("dpg/x", (B, 4, 80, 561), seed=1))
    y = torch.from_numpy(detgen.normal("dpg/y", (B,), seed=1))
    return x, y


def _worker(rank, world, port, use_graph, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from sitk import engine
    x, y = _data()
    shard = slice(rank * B // world, (rank + 1) * B // world)
    eng = engine.TrainEngine(_make_model("f32"), B // world, input_layout="patched", lr=LR, momentum=0.9,
                             process_group=dist.group.WORLD, bwd_slices=3, use_graph=use_graph, device="cuda:0")
    for _ in range(STEPS):
        eng.step(x[shard].cuda(), y[shard].cuda())
    torch.cuda.synchronize()
    if rank == 0:
        q.put(eng.fp.flat.cpu().numpy())      # by value: a tensor travels as an fd the parent must fetch while this process lives
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_rank_engine_matches_single_process_full_batch(use_graph):
    import sitk  # noqa: F401
    from sitk import engine
    x, y = _data()
    ref = engine.TrainEngine(_make_model("f32"), B, input_layout="patched", lr=LR, momentum=0.9, use_graph=False,
                             device="cuda:0")
    for _ in range(STEPS):
        ref.step(x.cuda(), y.cuda())
    want = ref.fp.flat.cpu()

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, use_graph, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = torch.from_numpy(q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    err = float((got.double() - want.double()).norm() / want.double().norm())
    assert err < 1e-6, err


def test_bench_gpus_2_as_typed_prints_one_json_line():
    """`python bench.py --gpus 2 ...` with no launcher around it: the script starts torch.distributed.run itself (child
    process) and rank 0's JSON line comes back through it (gloo transport: one GPU on this box)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-cpu-baseline",
                        "--no-probe", "--steps", "3", "--warmup", "2", "--batch", "4"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["value"] > 0
