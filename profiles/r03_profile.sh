#!/bin/bash
# Round-3 profiles, run on the GPU box from the repo root:  bash profiles/r03_profile.sh
# (1) rocprofv3 --kernel-trace --stats of the bench commands (per-kernel average durations),
# (2) HBM traffic per launch: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (never combined with a trace),
# (3) SQ counters of the headline stepper (profiles/pmc_run.sh).
# Everything lands under gpurun_out/r03prof/ ; profiles/r03_collect.py turns it into the files kept in profiles/
# (r03_kernel_stats_*.csv, r03_bench_*.json, r03_traffic_pmc.json) and prints the entries for profiles/traffic.json.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03prof
mkdir -p "$OUT"
stats() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d "$OUT/trace_$name" -o run --output-format csv -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || { echo "trace $name failed"; tail -3 "$OUT/bench_$name.err"; }
  find "$OUT/trace_$name" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_$name.csv" \;
  echo "stats $name done"
}
pmc() {     # name, counter, bench args...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr -d "$OUT/pmc_${name}_$ctr" -o run --output-format csv -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/pmc_${name}_$ctr.json" 2> "$OUT/pmc_${name}_$ctr.err" || { echo "pmc $name $ctr failed"; tail -3 "$OUT/pmc_${name}_$ctr.err"; }
  echo "pmc $name $ctr done"
}
WHAT=${1:-all}
if [ "$WHAT" = all ] || [ "$WHAT" = stats ]; then
  stats config3
  stats config3_20 --steps 20 --warmup 5
  stats config5 --config config5
  stats config5_20 --config config5 --steps 20 --warmup 5
  stats config5_total --config config5 --scaling strong --steps 200 --warmup 100
  stats config4 --config config4
  stats config2 --config config2
  stats config1 --config config1
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    pmc config3 $c --repeats 1
    pmc config5 $c --config config5 --repeats 1
    pmc config4 $c --config config4 --repeats 1
    pmc config2 $c --config config2 --repeats 1
  done
fi
if [ "$WHAT" = all ] || [ "$WHAT" = sq ]; then
  WALK=8 bash profiles/pmc_run.sh r03 > "$OUT/pmc_run_r03.log" 2>&1
  cp gpurun_out/pmc_r03/summary.json "$OUT/sq_config3.json" 2>/dev/null
fi
python3 profiles/r03_collect.py "$OUT"
