"""Reads a rocprofv3 kernel-trace CSV and reports, for one kernel name pattern, how much of its execution
time overlaps other kernels (i.e. whether the side stream really runs concurrently)."""
import csv
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
sel = [r for r in rows if re.search(pat, r[2])]
oth = [r for r in rows if not re.search(pat, r[2])]
tot = ov = 0
j = 0
for s, e, n, q in sel[len(sel) // 2:len(sel) // 2 + 200]:
    tot += e - s
    for s2, e2, n2, q2 in oth:
        if e2 <= s:
            continue
        if s2 >= e:
            break
        ov += min(e, e2) - max(s, s2)
print(f"{pat}: {len(sel)} launches; sampled total {tot / 1e3:.1f} us, overlapped with other kernels {ov / 1e3:.1f} us ({100.0 * ov / max(tot, 1):.1f} %)")
qs = {}
for s, e, n, q in rows:
    qs.setdefault(q, 0)
    qs[q] += 1
print("launches per stream/queue:", qs)
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _, _ in rows)
print(f"trace span {span / 1e6:.2f} ms, summed kernel time {busy / 1e6:.2f} ms")
