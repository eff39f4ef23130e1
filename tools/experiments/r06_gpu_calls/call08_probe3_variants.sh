#!/bin/bash
# round 6, call 8: does the cross-kernel corruption of k_static_ref_feat need (a) the missing wait state between `s_mov m0` and the LDS-DMA, (b) the function call in k_motion_mlp?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for t in m0nop nocall; do
  DYN_RAGGED=0 DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_$t.so timeout 600 python tools/concurrency_probe3.py > gpurun_out/r6c8_probe3_$t.txt 2>&1
  echo "== $t"; grep "trial" gpurun_out/r6c8_probe3_$t.txt | cut -c1-200
done
