#!/bin/bash
# HBM traffic of config 5's two kernels (feedback GEMM, lean stage kernel): FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
# (never combined with a trace), per launch; run on the GPU box from the repo root.  Output: gpurun_out/c5traffic/summary.json
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/c5traffic
mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$OUT/pmc_$c" -o run --output-format csv -- python3 bench.py --no-cpu-baseline --config config5 --steps 100 --warmup 20 --repeats 1 > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err" || { echo "pmc $c failed"; tail -3 "$OUT/bench_$c.err"; }
  echo "pmc $c done"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, f"pmc_{ctr}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            for key in ("crb_feedback_ws_kernel", "crb_stage_lean_kernel"):
                if key in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                    res.setdefault(key, {}).setdefault(ctr, []).append(float(row["Counter_Value"]))
summ = {}
for k, v in res.items():
    summ[k] = {c: {"mean_KiB": sum(x) / len(x), "min_KiB": min(x), "max_KiB": max(x), "dispatches": len(x)} for c, x in v.items()}
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:   # gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> x2
        summ[k]["hbm_bytes_per_launch"] = (2.0 * summ[k]["FETCH_SIZE"]["mean_KiB"] + summ[k]["WRITE_SIZE"]["mean_KiB"]) * 1024.0
json.dump(summ, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in summ.items()}))
PY
