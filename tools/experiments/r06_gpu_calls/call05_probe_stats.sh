#!/bin/bash
# round 6, call 5: statistics of the concurrency probe (24 trials of the three networks on fixed inputs while another chunk loops on a second stream): which network, how often,
# which points -- with the current library, without non-temporal accesses (nt0), with the round-5 library, and with ragged rows off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/concurrency_probe.py 8192 > gpurun_out/r6c5_probe_base.txt 2>&1; grep -v "^trial" gpurun_out/r6c5_probe_base.txt | tail -n 14 | cut -c1-420
DYN_RAGGED=0 timeout 900 python tools/concurrency_probe.py 8192 > gpurun_out/r6c5_probe_noragged.txt 2>&1; grep -v "^trial" gpurun_out/r6c5_probe_noragged.txt | tail -n 10 | cut -c1-420
DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_nt0.so timeout 900 python tools/concurrency_probe.py 8192 > gpurun_out/r6c5_probe_nt0.txt 2>&1; grep -v "^trial" gpurun_out/r6c5_probe_nt0.txt | tail -n 10 | cut -c1-420
DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_r5.so timeout 900 python tools/concurrency_probe.py 8192 > gpurun_out/r6c5_probe_r5.txt 2>&1; grep -v "^trial" gpurun_out/r6c5_probe_r5.txt | tail -n 10 | cut -c1-420
