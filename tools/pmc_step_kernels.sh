#!/bin/bash
# PMC passes over the three kernels of the BASELINE config-2 step, side by side (round-5 review task 4: name the dQ kernel's 33 %
# instruction wait).  Usage (on the GPU box, through gpurun): tools/pmc_step_kernels.sh TAG  -> gpurun_out/pmc_step_TAG/summary.txt
set -u
TAG=${1:-x}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_step_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU" \
           "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_FLAT SQ_THREAD_CYCLES_VALU" ; do
  i=$((i+1))
  for K in fwd dq dkdv; do
    timeout 120 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/p${i}_$K -o p -- python $REPO/tools/prof_shape.py $K 8 4096 > $OUT/p${i}_$K.log 2>&1
  done
done
cd $REPO
python - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fa::fa_" not in k or "preprocess" in k: continue
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k, d in per.items():
        for c, v in d.items():
            acc[k.split("<")[0].replace("void fa::", "")][c].append(v / max(1, len(disp[k])))
kernels = sorted(acc)
names = sorted({c for k in kernels for c in acc[k]})
print("%-34s" % "counter (per launch)" + "".join("%26s" % k[:25] for k in kernels))
for c in names:
    print("%-34s" % c + "".join("%26.4g" % (sum(acc[k][c]) / len(acc[k][c]) if acc[k][c] else float("nan")) for k in kernels))
def g(k, c):
    v = acc[k].get(c); return sum(v) / len(v) if v else float("nan")
print()
print("shares of SQ_WAVE_CYCLES")
for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
          "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM", "SQ_INST_CYCLES_SALU", "SQ_INST_CYCLES_VALU"):
    print("%-34s" % c + "".join("%25.1f%%" % (100 * g(k, c) / g(k, "SQ_WAVE_CYCLES")) for k in kernels))
print("per MFMA")
for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM"):
    print("%-34s" % c + "".join("%26.2f" % (g(k, c) / g(k, "SQ_INSTS_MFMA")) for k in kernels))
PY
find $OUT -name '*.csv' -delete; find $OUT -name '*.db' -delete
cat $OUT/summary.txt
