#!/bin/bash
# Which HIP API call is behind the __amd_rocclr_copyBuffer dispatches of a step?  (kernel trace joined with the HIP trace by correlation id)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/cbprobe; rm -rf $out
timeout -k 10 300 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $out -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-probe --no-graph > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
kt = glob.glob(out + "/*/*kernel_trace.csv")[0]
ht = glob.glob(out + "/*/*hip_api_trace.csv")[0]
api = {}
for r in csv.DictReader(open(ht)):
    api[r["Correlation_Id"]] = r["Function"]
rows = list(csv.DictReader(open(kt)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cnt = collections.Counter()
for i, r in enumerate(rows):
    if "copyBuffer" in r["Kernel_Name"]:
        prev = rows[i - 1]["Kernel_Name"][:50] if i else ""
        nxt = rows[i + 1]["Kernel_Name"][:50] if i + 1 < len(rows) else ""
        cnt[(api.get(r["Correlation_Id"], "?"), prev, nxt)] += 1
for k, v in cnt.most_common(30):
    print(v, k)
PY
rm -rf $out
