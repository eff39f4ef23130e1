export TMPDIR=/tmp; cd /tmp; R=/root/repo; O=$R/gpurun_out
B="python $R/bench.py --steps 10 --warmup 2 --cpu-rays 0 --no-x6"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_r1i -o bench -- $B > $O/prof_r1i.log 2>&1; echo stats rc=$?
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_r1i_f -o bench -- $B > $O/pmc_r1i_f.log 2>&1; echo fetch rc=$?
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_r1i_w -o bench -- $B > $O/pmc_r1i_w.log 2>&1; echo write rc=$?
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_r1i_s -o bench -- $B > $O/pmc_r1i_s.log 2>&1; echo sq rc=$?
cd $R; timeout 300 python tools/framebench.py > $O/frame_r1i.txt 2>&1; echo frame rc=$?; tail -3 $O/frame_r1i.txt
