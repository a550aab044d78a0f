#!/bin/bash
# PMC passes over BASELINE config 3's BACKWARD kernels (tools/prof_cfg3_bwd.py).  Usage: pmc_cfg3_bwd.sh TAG
set -u
TAG=${1:-x}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_cfg3_bwd_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $REPO/tools/prof_cfg3_bwd.py 20 > $OUT/trace.log 2>&1
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/p$i -o p -- python $REPO/tools/prof_cfg3_bwd.py > $OUT/p$i.log 2>&1
done
cd $REPO
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)):
    print(open(f).read())
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:90]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "fa_bwd" not in k and "bwd_pre" not in k: continue
        for c, v in d.items():
            print(f"{k[:60]:60s} {c:30s} {v / n[(k, c)]:.4e} per dispatch ({n[(k, c)]})")
PY
find $OUT -name '*.db' -delete; find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete
