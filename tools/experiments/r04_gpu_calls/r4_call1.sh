# round 4, GPU call 1: the full GPU suite, then k_motion_mlp old (m0) vs new (base) inside a frame, then the matrix pipe counters of the frame
cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/r4c1_tests.txt 2>&1; echo tests rc=$? 
tail -4 $O/r4c1_tests.txt
timeout 900 python tools/abbench.py --frame --rounds 2 --iters 20 base m0 > $O/r4c1_ab.txt 2>&1; echo ab rc=$?
tail -12 $O/r4c1_ab.txt
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU -d /root/repo/$O/r4c1_pmc -o frame -- python /root/repo/tools/framebench.py --frames 1 > /root/repo/$O/r4c1_pmc.log 2>&1; echo pmc rc=$?
cd /root/repo; python tools/rocpd_summary.py pmc $(find $O/r4c1_pmc -name '*results.db' | head -1) 2>/dev/null | grep -i "motion\|net_points\|static_views\|dynamic_views" | head -40
