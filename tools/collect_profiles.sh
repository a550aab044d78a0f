#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's numbers (run on the GPU box through gpurun):
#   1. --kernel-trace --stats of the very bench.py command            -> kernel durations
#   2. separate --pmc passes over a few steps of the same workload    -> MFMA/VALU/wait, LDS, HBM bytes
# Everything lands in gpurun_out/prof_$TAG ; tools/rocpd_summary.py turns it into the text under profiles/.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $OUT/trace -o t -- python $REPO/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
i=0
for PMC in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc$i -o p -- python $REPO/tools/prof_step.py 2 > $OUT/pmc$i.log 2>&1
done
cd $REPO
python tools/rocpd_summary.py --traffic-json $OUT/traffic.json $(find $OUT -name '*.db' | sort) > $OUT/summary.txt 2>&1
find $OUT -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
# keep the merge-back small: the raw databases stay on the box
find $OUT -name '*.db' -delete
find $OUT -name '*_kernel_trace.csv' -delete
ls -la $OUT
