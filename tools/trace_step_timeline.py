"""One training step of a rocprofv3 kernel-trace CSV as a timeline: for the last complete step (delimited by the optimizer
kernel), start / end (us from the step's first kernel) of every kernel whose name matches the pattern, per queue, and the
main chain's span -- to see WHEN side-stream launches really ran."""
import csv
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
opt = [i for i, r in enumerate(rows) if "sgd_dev_kernel" in r[2] or "adam_dev_kernel" in r[2]]
a, b = opt[-3] + 1, opt[-2] + 1
step = rows[a:b]
t0 = step[0][0]
print(f"step: {len(step)} kernels, {(step[-1][1] - t0) / 1e3:.1f} us")
for s, e, n, q in step:
    if re.search(pat, n):
        print(f"  q{q} {(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f} us)  {n[:60]}")
