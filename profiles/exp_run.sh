#!/bin/bash
# kernel-tuning helper: runs the default bench line once per library variant given
# (files under continuum-robot_amd/continuum_robot/_lib/, built with `make fast EXTRA=...`).
# usage: [BENCH_ARGS="--steps 20 --warmup 5"] bash profiles/exp_run.sh libA.so libB.so ...
mkdir -p gpurun_out
for lib in "$@"; do
  CRB_LIB_PATH=$PWD/continuum-robot_amd/continuum_robot/_lib/$lib python3 bench.py --no-cpu-baseline $BENCH_ARGS 2>gpurun_out/exp_$lib.err | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '%.4e'%d['value'], 'us/step %.2f'%(d['ms_per_step']*1e3), 'launch_ms %.4f'%d['roofline']['avg_launch_ms'], 'err', d['check'].get('rel_err_vs_oracle_last_beam'), 'blocks', {k: float('%.2g' % v) for k, v in d['check'].get('block_err_vs_oracle_last_beam', {}).items()})" || { echo "$lib FAILED"; tail -3 gpurun_out/exp_$lib.err; }
done
