#!/bin/bash
# round 5, call 17: k_motion_mlp launched only over the samples that keep their coefficients (the reference zeroes the last round(0.1 S) of every ray)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python tools/abbench.py --frame --iters 10 --rounds 2 base m1 > gpurun_out/r5c17_ab.txt 2>&1
cat gpurun_out/r5c17_ab.txt
DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_m1.so timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "motion or render_rays or render_single or full_frames or full_size or virtual or checkpoint" > gpurun_out/r5c17_parity.txt 2>&1; tail -4 gpurun_out/r5c17_parity.txt
