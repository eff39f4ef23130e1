#!/bin/bash
# round 6, call 9: the co-runner x victim matrix of the cross-stream corruption, shipped library and the variant whose one-wave-per-SIMD kernels claim the whole register file
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
DYN_RAGGED=0 timeout 900 python tools/concurrency_probe4.py > gpurun_out/r6c9_probe4_base.txt 2>&1; tail -n 6 gpurun_out/r6c9_probe4_base.txt | cut -c1-500
DYN_RAGGED=0 DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_x512.so timeout 900 python tools/concurrency_probe4.py > gpurun_out/r6c9_probe4_x512.txt 2>&1; tail -n 6 gpurun_out/r6c9_probe4_x512.txt | cut -c1-500
