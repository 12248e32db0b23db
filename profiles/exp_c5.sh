#!/bin/bash
# kernel-tuning helper: config-5 bench line per library variant (see exp_run.sh)
for lib in "$@"; do
  CRB_LIB_PATH=$PWD/continuum-robot_amd/continuum_robot/_lib/$lib python3 bench.py --no-cpu-baseline --config config5 --steps 400 2>gpurun_out/exp5_$lib.err | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '%.4e'%d['value'], 'us/step %.2f'%(d['ms_per_step']*1e3), 'err', d['check'].get('rel_err_vs_oracle_last_beam'))" || { echo "$lib FAILED"; tail -3 gpurun_out/exp5_$lib.err; }
done
