"""On WHICH stream does a c10d collective's kernel run when it is issued the engine's way?  (round 6, review item 1)

A one-rank RCCL group's in-place all-reduce(SUM) launches nothing, so `bench.py --dp-form` cannot tell.  Two one-rank operations
do reach the device: all_reduce(AVG) on floats (RCCL runs its one-rank pre-multiply kernel) and an out-of-place
all_gather_into_tensor (a device-to-device copy).  This tool issues both, three ways each, between MARKER kernels whose grid size
names the stream they were launched on (sitk_debug_occupy of the diagnostic build: 3 workgroups = the main stream, 5 = the
engine's bucket stream, 7 = a stream of the tool's own):

    A  `with torch.cuda.stream(bucket): dist.op(..., async_op=False)`     <- TrainEngine(dp_collective="stream"), the default
    B  `with torch.cuda.stream(bucket): dist.op(..., async_op=True)`      <- rounds 2 - 5 / dp_collective="group"
    C  `dist.op(..., async_op=False)` on the main stream                  <- the final bucket

Run under `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python tools/dp_collective_stream.py`, then
`python tools/dp_collective_stream.py --read DIR`: for every dispatch / copy between the markers it prints the trace's queue and stream
ids next to the markers' (tools/gpu_r6_dp.sh does both).  Needs SITK_LIB=.../libsitk_ab.so.
"""
import argparse
import csv
import ctypes
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read(out):
    rows = []
    for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), "kernel", r.get("Kernel_Name", "?"), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"),
                         r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Thread_Id", "?")))
    for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), "copy", r.get("Direction", "?"), r.get("Queue_Id", "-"), r.get("Stream_Id", "?"), "-",
                         r.get("Thread_Id", "?")))
    rows.sort()
    names = {"768": "MARKER main stream", "1280": "MARKER bucket stream", "1792": "MARKER tool's own stream"}
    started = False
    t0 = None
    for t, kind, name, q, s, grid, tid in rows:
        mark = names.get(str(grid)) if "debug_occupy" in name else None
        if mark:
            started = True
        if not started:
            continue
        if t0 is None:
            t0 = t
        label = mark or (name[:70])
        print(f"  {(t - t0) / 1e3:10.1f} us  {kind:6s} queue {q:>3s}  stream {s:>3s}  host thread {tid:>8s}  {label}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--read", default=None, help="directory of a rocprofv3 run of this tool: print where every operation landed")
    a = ap.parse_args()
    if a.read:
        read(a.read)
        return
    import torch
    import torch.distributed as dist
    import sitk  # noqa: F401
    from sitk import runtime as rt
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    lib = ctypes.CDLL(rt.LIB_PATH)
    lib.sitk_debug_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.sitk_debug_occupy.restype = ctypes.c_int
    bucket, own = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    main_s = torch.cuda.current_stream(dev)
    x = torch.randn(1 << 20, device=dev)
    y = torch.empty_like(x)
    print(f"torch {torch.__version__}; main stream {main_s.cuda_stream:#x}, bucket stream {bucket.cuda_stream:#x}, own {own.cuda_stream:#x}")

    def marker(n, s):
        assert lib.sitk_debug_occupy(n, 1, s.cuda_stream) == 0

    def ops(async_op):
        w = [dist.all_reduce(x, op=dist.ReduceOp.AVG, async_op=async_op), dist.all_gather_into_tensor(y, x, async_op=async_op)]
        for h in w:
            if h is not None:
                h.wait()

    for rep in range(2):          # (the first round creates communicators / streams; read the second)
        torch.cuda.synchronize()
        marker(3, main_s); marker(5, bucket); marker(7, own)
        torch.cuda.synchronize()
        print("A: synchronous, bucket stream current")
        marker(5, bucket)
        with torch.cuda.stream(bucket):
            ops(False)
        marker(5, bucket)
        torch.cuda.synchronize()
        print("B: async_op=True, bucket stream current")
        marker(5, bucket)
        with torch.cuda.stream(bucket):
            ops(True)
        marker(5, bucket)
        torch.cuda.synchronize()
        print("C: synchronous, main stream current")
        marker(3, main_s)
        ops(False)
        marker(3, main_s)
        torch.cuda.synchronize()
    assert torch.equal(x, y)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
