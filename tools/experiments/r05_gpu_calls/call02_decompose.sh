#!/bin/bash
# round 5, call 2: timing-only decomposition of the two-wave layer loop (no MFMAs / no LDS A reads / no ring) + the 0.96 gradient margin root cause
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/abbench.py --iters 20 --rounds 2 r4 n1 e1 e3 e4 e13 e134 > gpurun_out/r5c2_ab.txt 2>&1
tail -n 9 gpurun_out/r5c2_ab.txt
timeout 600 python tools/grad_rootcause.py > gpurun_out/r5c2_grad_default.txt 2>&1
tail -n 40 gpurun_out/r5c2_grad_default.txt
DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_x6.so timeout 600 python tools/grad_rootcause.py > gpurun_out/r5c2_grad_x6.txt 2>&1
tail -n 30 gpurun_out/r5c2_grad_x6.txt
