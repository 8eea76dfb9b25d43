#!/bin/bash
# usage (on the GPU box): bash tools/gpu_pmc2.sh <tag> <kernel-name-regex> <python script> [args...]
# Separate --pmc passes (never mixed with tracing; the program itself follows `--`), summarised per kernel into gpurun_out/pmc_<tag>.txt
tag=$1; shift; kre=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done
python - "$out" "$kre" <<'PY'
import csv, glob, re, sys, collections
out, kre = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
names = set()
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if re.search(kre, r["Kernel_Name"]):
            names.add(r["Kernel_Name"][:100])
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open(out + ".txt", "w") as fh:
    fh.write("kernels: " + " | ".join(sorted(names)) + "\n")
    for k in sorted(acc):
        line = f"{k:28s} avg/dispatch = {acc[k][0] / acc[k][1]:16.1f}   (n={acc[k][1]})"
        print(line); fh.write(line + "\n")
PY
rm -rf $out/p*/
