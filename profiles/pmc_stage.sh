#!/bin/bash
# HBM-side traffic and L1 behaviour of the config-5 kernels (stage kernel, feedback GEMM): one rocprofv3 --pmc pass per
# counter group, never combined with a trace.  usage (GPU box, repo root): bash profiles/pmc_stage.sh
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_stage
mkdir -p "$OUT"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_TA_BUSY_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp -d "$OUT/$tag" -o run --output-format csv -- python3 bench.py --no-cpu-baseline --config config5 --steps 40 --warmup 0 --repeats 1 > "$OUT/$tag.log" 2>&1 || { echo "$tag failed"; tail -3 "$OUT/$tag.log"; }
done
python3 - "$OUT" <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        name = "stage" if "crb_stage_lean_kernel" in k else ("gemm" if "crb_feedback" in k else None)
        if name:
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, d in acc.items():
    print(name, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "dispatches", {c: len(v) for c, v in d.items()})
P
