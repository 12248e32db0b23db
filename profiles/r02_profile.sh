#!/bin/bash
# Round-2 profiles, run on the GPU box from the repo root:  bash profiles/r02_profile.sh
# (1) rocprofv3 --kernel-trace --stats of the bench commands (per-kernel average durations),
# (2) HBM traffic of the stepper launches: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (never combined with a trace),
# (3) SQ counters of the headline stepper (profiles/pmc_run.sh).
# Everything lands under gpurun_out/r02prof/ ; profiles/r02_collect.py turns it into the files kept in profiles/.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r02prof
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
stats config3
stats config3_20 --steps 20 --warmup 5
stats config4 --config config4
stats config2 --config config2
stats config5 --config config5 --steps 200 --warmup 100
stats config1 --config config1
for c in FETCH_SIZE WRITE_SIZE; do
  pmc config3 $c --repeats 1
  pmc config4 $c --config config4 --repeats 1
  pmc config2 $c --config config2 --repeats 1
done
pmc config3 "TCC_HIT_sum TCC_MISS_sum" --repeats 1
WALK=8 bash profiles/pmc_run.sh r02 > "$OUT/pmc_run_r02.log" 2>&1
cp gpurun_out/pmc_r02/summary.json "$OUT/sq_config3.json" 2>/dev/null
WALK=4 bash profiles/pmc_run.sh r02f32 --config config4 > "$OUT/pmc_run_r02f32.log" 2>&1
cp gpurun_out/pmc_r02f32/summary.json "$OUT/sq_config4.json" 2>/dev/null
python3 profiles/r02_collect.py "$OUT"
