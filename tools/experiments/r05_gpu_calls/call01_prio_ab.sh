#!/bin/bash
# round 5, call 1: the issue-priority forms of the two-wave layer loop (B6_PRIO_MODE 0..4, dyn_mlp.h) + the one-asm operand split, against the round-4 library
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/abbench.py --frame --iters 20 --rounds 2 r4 p1 p0 p2 p3 p4 > gpurun_out/r5c1_ab.txt 2>&1
tail -n 9 gpurun_out/r5c1_ab.txt
DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_p1.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mlp_engine or static_pass or static_net or trained_scale" > gpurun_out/r5c1_parity_p1.txt 2>&1
tail -n 15 gpurun_out/r5c1_parity_p1.txt
