cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/r4c4_tests.txt 2>&1; echo tests rc=$?
tail -3 $O/r4c4_tests.txt
timeout 600 python tools/motionbench.py --rounds 2 m0 base mg0 mg8 vaddr nomfma > $O/r4c4_motion.txt 2>&1; echo mb rc=$?
tail -8 $O/r4c4_motion.txt
for tag in base nomfma; do
  lib=dynibar_amd/csrc/libdynibar_hip_$tag.so; [ $tag = base ] && lib=dynibar_amd/csrc/libdynibar_hip.so
  (DYNIBAR_HIP_LIB=$PWD/$lib python tools/motion_loop.py 8 > $O/r4c4_loop_$tag.txt 2>&1 &)
  sleep 4
  for i in 1 2 3; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|junction" | tr '\n' ';'; echo; sleep 1; done > $O/r4c4_smi_$tag.txt 2>&1
  sleep 5; cat $O/r4c4_loop_$tag.txt; cat $O/r4c4_smi_$tag.txt
done
export TMPDIR=/tmp; cd /tmp
DYNIBAR_HIP_LIB=/root/repo/dynibar_amd/csrc/libdynibar_hip.so timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES -d /root/repo/$O/r4c4_pmc_base -o m -- python /root/repo/tools/motion_once.py > /root/repo/$O/r4c4_pmc_base.log 2>&1
python /root/repo/tools/rocpd_summary.py pmc $(find /root/repo/$O/r4c4_pmc_base -name '*results.db' | head -1) 2>/dev/null | grep motion
