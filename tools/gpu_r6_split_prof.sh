#!/bin/bash
# round 6: kernel table + per-queue overlap of config 3 in its split-batch form (graph replays are traced kernel by kernel)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r6_cfg3_split
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r6_cfg3_split -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-probe --no-also --model small --patches 1280 --batch 32 > gpurun_out/prof_r6_cfg3_split.log 2>&1 || { tail -5 gpurun_out/prof_r6_cfg3_split.log; exit 1; }
f=$(ls gpurun_out/prof_r6_cfg3_split/*/*_kernel_stats.csv | head -1)
python tools/prof_summary.py "$f" --steps 23 --title "bench.py --model small --patches 1280 --batch 32 in its split-batch form (two concurrent half-batch hipGraph steps; rocprofv3 --kernel-trace --stats)" > gpurun_out/prof_r6_cfg3_split.md
t=$(ls gpurun_out/prof_r6_cfg3_split/*/*_kernel_trace.csv | head -1)
python - "$t" >> gpurun_out/prof_r6_cfg3_split.md <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "sitk_spin" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete step: between the last two sgd_dev_kernel launches
opt = [i for i, r in enumerate(rows) if "sgd_dev_kernel" in r["Kernel_Name"]]
a, b = opt[-2] + 1, opt[-1] + 1
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
per_q = collections.defaultdict(lambda: [0, 0])
ev = []
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r["Queue_Id"]
    per_q[q][0] += e - s
    per_q[q][1] += 1
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = {0: 0, 1: 0, 2: 0}
n, last = 0, t0
for t, d in ev:
    busy[min(n, 2)] += t - last
    n += d
    last = t
print(f"\nlast complete step of the trace: {(t1 - t0) / 1e3:.0f} us, {len(step)} kernels")
for q, (ns, k) in sorted(per_q.items()):
    print(f"  queue {q}: {k} kernels, {ns / 1e3:.0f} us of kernel time")
tot = t1 - t0
print(f"  wall time with 0 / 1 / >= 2 kernels in flight: {busy[0] / tot:.1%} / {busy[1] / tot:.1%} / {busy[2] / tot:.1%}")
PY
rm -f gpurun_out/prof_r6_cfg3_split/*/*_kernel_trace.csv
cat gpurun_out/prof_r6_cfg3_split.md | cut -c1-170 | tail -32
