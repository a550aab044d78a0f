#!/bin/bash
# PMC passes over the forward only (tools/prof_fwd.py [--nc]); prints per-kernel counter sums.  Usage: pmc_fwd.sh TAG [--nc]
set -u
TAG=${1:-x}; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/p$i -o p -- python $REPO/tools/prof_fwd.py "$@" > $OUT/p$i.log 2>&1
done
cd $REPO
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "fa_fwd" not in k: continue
        print(k)
        for c, v in d.items():
            print(f"   {c:36s} {v / n[(k, c)]:.4e} per dispatch ({n[(k, c)]} dispatches)")
PY
find $OUT -name '*.db' -delete; find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete
