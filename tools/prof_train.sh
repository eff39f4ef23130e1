#!/bin/bash
# Round profiles of the training iteration (run on the GPU box through gpurun): rocprofv3 kernel stats and HBM-traffic PMC passes of
# tools/trainbench.py <rays> full.  Summaries land in gpurun_out/prof_train/; copy the ones to be judged into profiles/.
set -u
RAYS=${1:-3072}
R=$PWD
OUT=$R/gpurun_out/prof_train
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $R/tools/trainbench.py $RAYS full"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o tb -- $CMD > $OUT/stats.log 2>&1
python $R/tools/rocpd_summary.py stats $(find $OUT/stats -name '*.db' | head -1) > $OUT/train_full_kernel_stats.txt 2>&1
DBS=""
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$N -o tb -- $CMD > $OUT/pmc_$N.log 2>&1
  DBS="$DBS $(find $OUT/pmc_$N -name '*.db' | head -1)"
done
python $R/tools/rocpd_summary.py pmc $DBS > $OUT/train_full_pmc.txt 2>&1
cd $R
rm -rf $OUT/stats $OUT/pmc_*   # the databases are large; the text summaries are what gets committed
head -30 $OUT/train_full_kernel_stats.txt | cut -c1-140
grep -E "k_train_gemm|k_train_act_bwd|k_gather_bwd" $OUT/train_full_pmc.txt | head -40 | cut -c1-150
