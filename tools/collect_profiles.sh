#!/bin/bash
# Round profiles (run on the GPU box through gpurun): rocprofv3 kernel stats and PMC passes of the bench command and of one full frame.
# Summaries land in gpurun_out/prof_<tag>/*.txt|json; copy the ones to be judged into profiles/.
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 10 --warmup 2 --cpu-rays 0 --no-extra --no-traffic --no-x6"
FRAME="python $PWD/tools/framebench.py --frames 1"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/bench_stats -- $BENCH > $OUT/bench_stats.log 2>&1
python $OLDPWD/tools/rocpd_summary.py stats $(find $OUT/bench_stats -name '*.db' | head -1) > $OUT/${TAG}_bench_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/frame_stats -- $FRAME > $OUT/frame_stats.log 2>&1
python $OLDPWD/tools/rocpd_summary.py stats $(find $OUT/frame_stats -name '*.db' | head -1) > $OUT/${TAG}_frame_kernel_stats.txt 2>&1
DBS_B=""; DBS_F=""
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C -d $OUT/bench_pmc_$N -- $BENCH > $OUT/bench_pmc_$N.log 2>&1
  DBS_B="$DBS_B $(find $OUT/bench_pmc_$N -name '*.db' | head -1)"
done
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C -d $OUT/frame_pmc_$N -- $FRAME > $OUT/frame_pmc_$N.log 2>&1
  DBS_F="$DBS_F $(find $OUT/frame_pmc_$N -name '*.db' | head -1)"
done
python $OLDPWD/tools/rocpd_summary.py pmc $DBS_B > $OUT/${TAG}_bench_pmc.txt 2>&1
python $OLDPWD/tools/rocpd_summary.py pmc $DBS_F > $OUT/${TAG}_frame_pmc.txt 2>&1
python $OLDPWD/tools/rocpd_summary.py traffic $OUT/${TAG}_traffic.json $DBS_B > /dev/null 2>&1
python $OLDPWD/tools/rocpd_summary.py traffic $OUT/${TAG}_frame_traffic.json $DBS_F > /dev/null 2>&1
cd $OLDPWD
python tools/framebench.py > $OUT/${TAG}_frame_nvi_288x512.txt 2>&1
python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/bench_n1.err
# keep only the summaries (the databases are large)
find $OUT -name '*.db' -delete; find $OUT -type d -empty -delete
ls -la $OUT
