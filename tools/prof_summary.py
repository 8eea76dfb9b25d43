"""Condense a rocprofv3 --kernel-trace --stats run (CSV output) into a small table for profiles/.

    python tools/prof_summary.py gpurun_out/prof/<host>/<pid>_kernel_stats.csv --steps 23 > profiles/r01_xxx.md
"""
import argparse
import csv
import glob
import re
import sys


def short(name):
    name = name.replace("sitk::", "")
    m = re.match(r"_ZN4sitk(\d+)([A-Za-z_0-9]+)", name)
    if m:
        n = int(m.group(1))
        rest = name[len("_ZN4sitk") + len(m.group(1)):]
        base, targs = rest[:n], rest[n:]
        targs = (targs.replace("DF16b", "bf16,").replace("IfffL", "I f32,f32,f32,L").replace("ELi", ",").replace("Li", "")
                 .replace("EEvNS_10GemmParamsE", "").replace("EEvNS_11WgradParamsE", ""))
        return f"{base}<{targs[:40]}>"
    return name[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, default=0, help="train steps in the profiled run (for per-step columns)")
    ap.add_argument("--title", default="rocprofv3 --kernel-trace --stats")
    a = ap.parse_args()
    path = a.csv
    if "*" in path:
        path = sorted(glob.glob(path))[-1]
    rows = list(csv.DictReader(open(path)))
    # (the engine's stream placement probe runs at construction: its spin kernels are not part of any step)
    probe = [r for r in rows if "sitk_spin_kernel" in r["Name"]]
    rows = [r for r in rows if "sitk_spin_kernel" not in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {a.title}\n")
    print(f"source: `{path}`; total kernel time {tot / 1e6:.2f} ms" + (f" over {a.steps} steps = {tot / 1e6 / a.steps:.3f} ms/step" if a.steps else ""))
    if probe:
        print(f"(left out: {sum(int(r['Calls']) for r in probe)} `sitk_spin_kernel` launches of the stream placement probe at engine "
              f"construction, {sum(float(r['TotalDurationNs']) for r in probe) / 1e6:.2f} ms in all)")
    print("\n| % | calls | avg us | min us | max us | kernel |\n|---:|---:|---:|---:|---:|---|")
    for r in rows:
        pct = float(r["TotalDurationNs"]) / tot * 100
        if pct < 0.05:
            continue
        print(f"| {pct:.1f} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
              f"{float(r['MaxNs']) / 1e3:.1f} | `{short(r['Name'])}` |")


if __name__ == "__main__":
    sys.exit(main())
