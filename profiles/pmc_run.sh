#!/bin/bash
# Collects SQ counters for the stepper kernel: one rocprofv3 --pmc pass per counter group
# (never combined with trace domains), ONE launch of 200 fused RK4 steps.
# usage (on the GPU box, from the repo root):  [WALK=8] bash profiles/pmc_run.sh TAG [bench.py args...]
# WALK = beams one launched wave walks over (config 3: 4096 beams on 512 resident workgroups = 8; config 4: 4; default 1)
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d "$OUT/g$i" -o run --output-format csv -- python3 bench.py --steps 200 --launch-steps 200 --warmup 0 --no-cpu-baseline "$@" > "$OUT/g$i.log" 2>&1 || { tail -5 "$OUT/g$i.log"; echo "group $i failed"; }
done
python3 profiles/pmc_collect.py "$OUT" "$TAG" "${WALK:-1}"
