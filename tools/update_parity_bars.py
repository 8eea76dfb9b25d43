"""Copy the bf16 errors a GPU run of the tests measured (gpurun_out/parity_measured.json, written by
tests/conftest.py) into the committed tests/golden/parity_measured_bf16.json, from which tests/parity_bars.py
derives the bf16 bars (2 x measured, capped by the fixed ceilings of tests/parity_bars.py).
Usage: python tools/update_parity_bars.py [--merge] [file]      --merge keeps entries the run did not produce.
ONE recording run: the bench path is bitwise reproducible since round 3, so there is no "largest of N runs" mode any more
(round 2's --max only ever raised bars)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "parity_measured.json")
dst = os.path.join(ROOT, "tests", "golden", "parity_measured_bf16.json")
files = [a for a in sys.argv[1:] if not a.startswith("--")] or [src]
keep = "--merge" in sys.argv and os.path.exists(dst)
out = json.load(open(dst)) if keep else {}
rec = {}
for f in files:
    for k, v in json.load(open(f)).items():
        rec[k] = v
for k, v in rec.items():
    if k.startswith("bf16/"):
        key = k[len("bf16/"):]
        v = float(f"{v:.3e}")
        out[key] = v
json.dump(dict(sorted(out.items())), open(dst, "w"), indent=1)
print(f"{len(out)} bf16 entries -> {dst}")
# the f16 margin guard of tests/parity_bars.py: every f16 value above 90 % of its fixed bar, as measured by this run
sys.path.insert(0, ROOT)
from tests import parity_bars as pb  # noqa: E402
watch = json.load(open(pb.WATCH_PATH)) if keep and os.path.exists(pb.WATCH_PATH) else {}
for k, v in rec.items():
    if k.startswith("f16/"):
        key = k[len("f16/"):]
        case, metric = key.rsplit("/", 1)
        b = pb.F16_EXCEPTIONS.get(key, 1e-3)
        if v > pb.F16_WATCH_FROM * b:
            watch[key] = float(f"{v:.3e}")
        elif key in watch:
            del watch[key]
json.dump(dict(sorted(watch.items())), open(pb.WATCH_PATH, "w"), indent=1)
print(f"{len(watch)} watched f16 entries -> {pb.WATCH_PATH}: {watch}")
f32 = {k: v for k, v in rec.items() if k.startswith("f32/")}
if f32:
    worst = max(f32.items(), key=lambda kv: kv[1])
    print(f"f32 entries: {len(f32)}, worst {worst[0]} = {worst[1]:.3e}")
