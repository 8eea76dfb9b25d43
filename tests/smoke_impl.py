"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0 checked against the CPU
oracle (test infrastructure; the only place outside tests/bench that may import oracle/)."""
import numpy as np
import torch


def run_smoke():
    from oracle import detgen, sit_oracle
    import sitk  # noqa: F401
    from sitk import engine, tables
    from sitk.models.sit import SiT

    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    dev = "cuda:0"
    kw = dict(sit_oracle.MODEL_SIZES["tiny"], depth=2, num_patches=320, num_vertices=153, num_channels=4)
    ref = sit_oracle.SiT(**kw)
    vals = detgen.fill_state_dict(ref.state_dict(), seed=1)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    xs = detgen.normal("smoke/x", (4, 40962, 4), seed=1)
    y = detgen.normal("smoke/y", (4,), seed=1)
    patched = sit_oracle.gather_patches(np.ascontiguousarray(xs.transpose(0, 2, 1)), tables.load_table(320, 153))
    lref = torch.nn.functional.mse_loss(ref(torch.from_numpy(patched)).squeeze(), torch.from_numpy(y))
    lref.backward()
    for dtype, tol in (("f32", 1e-3), ("f16", 2e-3), ("bf16", 1.5e-2)):   # bf16: 2-3 x the element-wise gradient error measured (5e-3)
        model = SiT(**kw, compute_dtype=dtype)
        model.load_state_dict(ref.state_dict())
        eng = engine.TrainEngine(model, 4, input_layout="surface", lr=0.0, momentum=0.0, use_graph=False, keep_grads=True)
        loss = float(eng.step(torch.from_numpy(xs).to(dev), torch.from_numpy(y).to(dev)))
        assert abs(loss - float(lref)) / float(lref) < tol, (dtype, loss, float(lref))
        worst = (0.0, "")
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            e = float((p.grad.cpu().double() - q.grad.double()).norm() / (q.grad.double().norm() + 1e-30))
            assert e < tol, (dtype, k, e)
            worst = max(worst, (e, k))
        print(f"smoke[{dtype}]: loss {loss:.6f} (oracle {float(lref):.6f}), worst gradient {worst[0]:.2e} ({worst[1]}; bar {tol:g})")
    torch.cuda.synchronize()
