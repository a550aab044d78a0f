#!/bin/bash
# PMC passes over the head-dim-256 forward (compiler-scheduled fa_fwd_kernel<., 256, ...>): what keeps its matrix pipe at ~44 % of the
# socket's ceiling?  Usage (GPU box): tools/pmc_fwd_d256.sh TAG [lib]  -> gpurun_out/pmc_fwd256_TAG/summary.txt
set -u
TAG=${1:-x}
REPO=$(pwd)
[ -n "${2:-}" ] && export FA_MI355_LIB=$REPO/$2
OUT=$REPO/gpurun_out/pmc_fwd256_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" ; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/p${i} -o p -- python $REPO/tools/prof_shape.py fwd 8 4096 8 256 > $OUT/p${i}.log 2>&1
done
cd $REPO
python - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(float); disp = set()
    for r in csv.DictReader(open(f)):
        if "fa::fa_fwd" not in r["Kernel_Name"]: continue
        per[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    for c, v in per.items():
        acc[c].append(v / max(1, len(disp)))
g = lambda c: sum(acc[c]) / len(acc[c]) if acc.get(c) else float("nan")
for c in sorted(acc):
    print("%-34s %14.5g" % (c, g(c)))
print()
for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
          "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VALU"):
    print("%-34s %6.1f %% of SQ_WAVE_CYCLES" % (c, 100 * g(c) / g("SQ_WAVE_CYCLES")))
print("SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs ...) raw ratio to SQ_BUSY_CYCLES: %.3f" % (g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")))
for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM"):
    print("%-34s %6.2f per MFMA" % (c, g(c) / g("SQ_INSTS_MFMA")))
PY
find $OUT -name '*.csv' -delete; find $OUT -name '*.db' -delete
cat $OUT/summary.txt
