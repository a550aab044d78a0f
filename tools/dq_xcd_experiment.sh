#!/bin/bash
# dQ kernel: paired (4 (b,h) units live per XCD, 8 workgroups each) vs unpaired blocks (2 units live, 16 workgroups each)
REPO=$(pwd)
for l in product dqnopair product dqnopair; do
  if [ $l = product ]; then python tools/bwd_perf.py 2>&1 | grep "^B" ; else FA_MI355_LIB=tools/variants/libfa_$l.so python tools/bwd_perf.py 2>&1 | grep "^B" | sed "s/^/[$l] /"; fi
done
cd /tmp && export TMPDIR=/tmp
for l in product dqnopair; do
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
    if [ $l = product ]; then unset FA_MI355_LIB; else export FA_MI355_LIB=$REPO/tools/variants/libfa_$l.so; fi
    rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $REPO/gpurun_out/dqxcd_$l -o p -- python $REPO/tools/prof_shape.py dq 8 4096 > /dev/null 2>&1
    python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$REPO/gpurun_out/dqxcd_$l/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dq_asm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c in acc: print("$l", c, "%.4e per dispatch (%d)" % (acc[c] / n[c], n[c]))
PY
    rm -rf $REPO/gpurun_out/dqxcd_$l
  done
done
