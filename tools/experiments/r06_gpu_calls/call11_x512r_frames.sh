#!/bin/bash
# round 6, call 11: six two-stream processes (cold + warm frame each) with the library in which the one-wave-per-SIMD kernels AND k_static_ref_feat claim the whole register file,
# all rays against one one-stream process; the same with the library in which only the one-wave-per-SIMD kernels do (x512)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for t in x512r x512; do
  DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_$t.so timeout 900 python tools/ragged_frame_ab.py --many > gpurun_out/r6c11_frame_ab_$t.txt 2>&1; echo "== $t"; tail -n 13 gpurun_out/r6c11_frame_ab_$t.txt | cut -c1-260
done
timeout 900 python tools/abbench.py --frame --iters 10 --rounds 2 x512 x512r > gpurun_out/r6c11_ab.txt 2>&1; tail -n 4 gpurun_out/r6c11_ab.txt | cut -c1-330
