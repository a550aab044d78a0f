#!/bin/bash
# HBM traffic of the config-4 decode kernels (FETCH_SIZE pass + kernel trace): tools/bench_decode.py --fp8 under rocprofv3.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_decode
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p1 -o p -- python $REPO/tools/bench_decode.py --fp8 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python $REPO/tools/bench_decode.py --fp8 > $OUT/t.log 2>&1
cd $REPO
python tools/rocpd_summary.py $(find $OUT/p1 -name '*.db' | sort) > $OUT/summary.txt 2>&1
find $OUT -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name '*.db' -delete; find $OUT -name '*_kernel_trace.csv' -delete
grep -E "decode|FETCH" $OUT/summary.txt | head -40
