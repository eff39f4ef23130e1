#!/bin/bash
# round 5, call 12: the blend with the next tile's loads in flight under the current tile's layers: 512 threads (two waves per SIMD; spills) and 256 threads (one per SIMD)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python tools/abbench.py --frame --iters 20 --rounds 2 base w512 w256 > gpurun_out/r5c12_ab.txt 2>&1
cat gpurun_out/r5c12_ab.txt
DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_w512.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "static_net or static_pass or segment_widths or trained" > gpurun_out/r5c12_parity.txt 2>&1; tail -3 gpurun_out/r5c12_parity.txt
