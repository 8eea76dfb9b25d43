"""Parity bars of the GPU model / engine tests.

f32 compute mode (v_mfma_f32_16x16x4_f32, exact fp32) is held to north_star's bar with margin: 2e-4 on
outputs and losses, 1e-3 on every gradient.

bf16 mode (the benchmarked dtype) cannot meet 1e-3 (one 2^-9 rounding per MFMA operand over up to 12
layers); its bar per case and metric is 2 x THE ERROR MEASURED ON AN MI355X (the largest of eight recording runs: float atomics
reorder the gradient sums, and one flipped bf16 rounding of a weight moves the multi-step engine metrics by up to 10 x
from run to run), recorded in
tests/golden/parity_measured_bf16.json (written by a GPU run of these tests: every `check()` call records
its value, tests/conftest.py dumps the records to gpurun_out/parity_measured.json at the end of the
session, and tools/update_parity_bars.py copies the bf16 entries into the committed file).  A case without
a recorded value fails -- a new case must be measured before it can pass -- unless SITK_PARITY_RECORD=1
(the recording run).  Every check prints `parity <case> <metric>: measured / bar`.
"""
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
MEASURED_PATH = os.path.join(_HERE, "golden", "parity_measured_bf16.json")
F32_BARS = {"out": 2e-4, "loss": 2e-4, "grad": 1e-3, "param": 1e-5}
FLOOR = 2e-4          # bars never go below this (measured errors of ~0 would make the bar meaningless)
ENGINE_FLOOR = 5e-4   # multi-step engine metrics: heavy-tailed from run to run (median 5e-5, 2e-4 once in eight runs)
RECORDS = {}

try:
    _MEASURED = json.load(open(MEASURED_PATH))
except (OSError, ValueError):
    _MEASURED = {}


def bar(case, metric, dtype, kind):
    """kind: one of F32_BARS' keys (which north-star bar applies in f32 mode)."""
    if dtype == "f32":
        return F32_BARS[kind]
    m = _MEASURED.get(f"{case}/{metric}")
    if m is None:
        return None
    return max(2.0 * m, ENGINE_FLOOR if case.startswith("engine/") else FLOOR)


def check(case, metric, dtype, value, kind):
    """Assert value <= bar; record it; print both."""
    value = float(value)
    RECORDS[f"{dtype}/{case}/{metric}"] = value
    b = bar(case, metric, dtype, kind)
    recording = os.environ.get("SITK_PARITY_RECORD") == "1"
    print(f"parity {dtype} {case} {metric}: measured {value:.3e} / bar {b if b is None else format(b, '.3e')}")
    if b is None:
        assert recording, (f"no measured bf16 error recorded for {case}/{metric}: run the GPU tests with "
                           f"SITK_PARITY_RECORD=1 and tools/update_parity_bars.py")
        return
    assert value <= b or recording, f"{dtype} {case} {metric}: {value:.3e} > bar {b:.3e}"
