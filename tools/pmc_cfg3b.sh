#!/bin/bash
# short PMC set over BASELINE config 3's forward: VALU / MFMA busy and their co-execution.  Usage: pmc_cfg3b.sh TAG
set -u
TAG=${1:-x}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_cfg3b_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/p$i -o p -- python $REPO/tools/prof_fwd.py --cfg3 > $OUT/p$i.log 2>&1
done
cd $REPO
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "fa_fwd" not in k: continue
        for c, v in d.items():
            print(f"{c:30s} {v / n[(k, c)]:.4e}")
PY
rm -rf $OUT/p*/
