"""Instruction mix of one kernel in a hipcc -S listing: python tools/isa_summary.py file.s <symbol substring> [--loop]"""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(sym) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
c = collections.Counter()
for l in body:
    t = l.strip().split(" ")[0] if l.strip() else ""
    if t.startswith(("v_", "s_waitcnt", "ds_", "global", "buffer", "s_barrier", "s_cbranch", "scratch", "s_nop")):
        c["v_mfma" if t.startswith("v_mfma") else t] += 1
print(len(body), "lines")
for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:45]:
    print(f"{v:6d} {k}")
print("vmcnt waits:", [l.strip() for l in body if "s_waitcnt" in l and "vmcnt" in l])
if "--dump" in sys.argv:
    print("\n".join(body))
