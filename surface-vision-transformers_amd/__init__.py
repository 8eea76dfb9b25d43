"""sitk -- MI355X (gfx950) native training hot path for the Surface Vision Transformer.

Layout:
  csrc/        HIP kernels + the C ABI (include/sitk.h) -> libsitk.so
  runtime.py   ctypes binding (fails loudly when the library is missing; no CPU fallback)
  ops.py       tensor-level wrappers
  models/      host-side mirror of the reference's models/sit.py and models/mpp.py
  engine.py    fused train step (gather -> fwd -> loss -> bwd -> optimizer), hipGraph capture, data-parallel gradient
               all-reduce per backward slice (RCCL via torch.distributed), resident data set + device-side lr state
  probe.py     per-kernel timing of one encoder layer for bench.py's `roofline` object
  tables.py    icosahedral patch-index tables (data/*.npy)
"""
__version__ = "0.1.0"
