cd /tmp; export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $O/r4c20_$N -o m -- python $R/tools/motion_once.py > $O/r4c20_$N.log 2>&1
  python $R/tools/rocpd_summary.py pmc $(find $O/r4c20_$N -name '*results.db' | head -1) 2>/dev/null | grep motion | cut -c70-160
done
# the bench step too (view / point / blend kernels)
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQC_TC_INST_REQ SQC_TC_STALL GRBM_GUI_ACTIVE"; do
  N=b_$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $O/r4c20_$N -o m -- python $R/bench.py --steps 6 --warmup 2 --cpu-rays 0 --no-extra --no-traffic > $O/r4c20_$N.log 2>&1
  python $R/tools/rocpd_summary.py pmc $(find $O/r4c20_$N -name '*results.db' | head -1) 2>/dev/null | grep "static_views\|net_points\|blend" | cut -c1-160
done
find $O -name '*results.db' -path '*r4c20*' -delete
