#!/bin/bash
# round 6, call 10: whole frames with the exclusive-CU library (one-wave-per-SIMD kernels claim the whole register file): reproducible on two and three chunk streams? what does it cost?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_x512.so timeout 900 python tools/ragged_frame_ab.py --quick > gpurun_out/r6c10_frame_ab_x512.txt 2>&1; tail -n 12 gpurun_out/r6c10_frame_ab_x512.txt | cut -c1-260
timeout 1500 python tools/abbench.py --frame --iters 20 --rounds 2 base x512 > gpurun_out/r6c10_ab.txt 2>&1; tail -n 8 gpurun_out/r6c10_ab.txt | cut -c1-330
DYNIBAR_CHUNK_STREAMS=1 timeout 900 python tools/abbench.py --frame --iters 5 --rounds 1 x512 > gpurun_out/r6c10_ab_s1.txt 2>&1; tail -n 3 gpurun_out/r6c10_ab_s1.txt | cut -c1-330
