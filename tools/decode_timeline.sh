#!/bin/bash
# Config 4 at H_k = 8 (fp8 and fp16 caches) and H_k = 32: (1) the kernel time line of a step (launch order, durations, gaps between
# kernels) from a rocprofv3 kernel trace, (2) FETCH_SIZE of every decode kernel in a separate counter pass.
# -> gpurun_out/decode_timeline/{timeline.txt, traffic.txt}   (profiles/r05_decode_traffic.txt)
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/decode_timeline
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $REPO/tools/cfg4_hk8.py 1 > $OUT/t.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p -o p -- python $REPO/tools/cfg4_hk8.py 1 > $OUT/p.log 2>&1
cd $REPO
python tools/decode_timeline.py $(find $OUT/t -name '*kernel_trace.csv' | head -1) > $OUT/timeline.txt 2>&1
python tools/decode_timeline.py --pmc $(find $OUT/p -name '*counter_collection.csv' | head -1) > $OUT/traffic.txt 2>&1
cat $OUT/t.log | grep "lib=" ; cat $OUT/timeline.txt | head -80; cat $OUT/traffic.txt | head -40
find $OUT -name '*.csv' -size +20M -delete
