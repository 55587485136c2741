#!/bin/bash
# Run on the GPU box (through gpurun).  Usage: tools/profile_gpu.sh TAG WORKLOAD [pmc]
# Writes gpurun_out/prof_TAG/{trace,pmc_*}/ ; summarise afterwards with tools/summarize_profile.py.
set -u
TAG=${1:-r01}; WL=${2:-4096_J40}; DO_PMC=${3:-pmc}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --workload $WL --steps 5 --warmup 2 --no-extra --no-cpu-baseline"
# the trace is of the bench command itself (default K/W; the CPU baseline and the extra
# workloads run too, they launch no additional kernels of the traced workload's names)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload $WL --no-extra > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
if [ "$DO_PMC" = "pmc" ]; then
  i=0
  for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
              "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE32_INSTS SQ_INSTS_BRANCH" \
              "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_$i -- python3 $ARGS > $OUT/pmc_$i.log 2>&1 || { echo "pmc pass $i failed ($CTRS)"; tail -3 $OUT/pmc_$i.log; }
  done
fi
echo "profile done: $OUT"
