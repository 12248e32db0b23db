#!/bin/bash
# SQ / cache counters of the feedback GEMM (profiles/exp_gemm.py); usage: bash profiles/pmc_gemm.sh TAG
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcg_$TAG
mkdir -p "$OUT"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d "$OUT/g$i" -o run --output-format csv -- python3 profiles/exp_gemm.py > "$OUT/g$i.log" 2>&1 || { tail -5 "$OUT/g$i.log"; echo "group $i failed"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
tot, n = {}, {}
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "crb_feedback" not in row["Kernel_Name"]:
            continue
        c = row["Counter_Name"]
        tot[c] = tot.get(c, 0.0) + float(row["Counter_Value"]); n[c] = n.get(c, 0) + 1
res = {c: tot[c] / n[c] for c in tot}   # per dispatch
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
