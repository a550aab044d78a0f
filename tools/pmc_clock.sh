#!/bin/bash
# GRBM_GUI_ACTIVE per kernel -> effective clock = count / kernel wall time (MI355X_MICROARCH.md "DVFS give-back").
# One counter pass, kernel trace only (no other trace domain).  Run on the GPU box:  tools/pmc_clock.sh r03
set -u
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out/clock_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/gui -o g -- python $REPO/tools/prof_step.py 6 > $OUT/gui.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/sq -o s -- python $REPO/tools/prof_step.py 6 > $OUT/sq.log 2>&1
cd $REPO
python tools/pmc_clock_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name '*.db' -delete
