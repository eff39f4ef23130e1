cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 600 python tools/motionbench.py --rounds 2 base nodma nodbl nomfma dmaonly ldsonly saddr > $O/r4c3_motion.txt 2>&1; echo mb rc=$?
tail -9 $O/r4c3_motion.txt
export TMPDIR=/tmp; cd /tmp
for tag in base nodma nodbl nomfma dmaonly; do
  lib=/root/repo/dynibar_amd/csrc/libdynibar_hip_$tag.so; [ $tag = base ] && lib=/root/repo/dynibar_amd/csrc/libdynibar_hip.so
  DYNIBAR_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES -d /root/repo/$O/r4c3_pmc_$tag -o m -- python /root/repo/tools/motion_once.py > /root/repo/$O/r4c3_pmc_$tag.log 2>&1
  echo "== $tag rc=$?"; python /root/repo/tools/rocpd_summary.py pmc $(find /root/repo/$O/r4c3_pmc_$tag -name '*results.db' | head -1) 2>/dev/null | grep motion
done
DYNIBAR_HIP_LIB=/root/repo/dynibar_amd/csrc/libdynibar_hip.so timeout 200 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum -d /root/repo/$O/r4c3_pmc_ta -o m -- python /root/repo/tools/motion_once.py > /root/repo/$O/r4c3_pmc_ta.log 2>&1
echo "== ta rc=$?"; python /root/repo/tools/rocpd_summary.py pmc $(find /root/repo/$O/r4c3_pmc_ta -name '*results.db' | head -1) 2>/dev/null | grep motion; tail -3 /root/repo/$O/r4c3_pmc_ta.log
