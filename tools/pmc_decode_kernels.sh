#!/bin/bash
# PMC passes over the decode kernels of config 4 (H_k 32: token-major kernel) and its H_k 8 variant (MFMA kernels), fp8 and fp16 caches:
# issue / wait split, instruction mix, MFMA busy.  -> gpurun_out/pmc_decode_kernels/summary.txt   (profiles/r05_decode_traffic.txt, section 6)
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_decode_kernels
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/p$i -o p -- python $REPO/tools/cfg4_hk8.py 1 > $OUT/p$i.log 2>&1
done
cd $REPO
python - > $OUT/summary.txt <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fa_decode" not in k: continue
        k = k.replace("void fa::", "").replace("(fa::DecArgs)", "").replace("fa::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
import statistics as st
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:30s} {st.median(v):.4e}   ({len(v)} dispatches)")
PY
cat $OUT/summary.txt
find $OUT -name '*.csv' -size +5M -delete
