#!/bin/bash
# round 5, call 8: the blend's colour from the row record (b1) and the dense view kernel with the view count at compile time (d1) against the current default
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/abbench.py --frame --iters 20 --rounds 2 base b1 d1 > gpurun_out/r5c8_ab.txt 2>&1
cat gpurun_out/r5c8_ab.txt
DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_d1.so timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not edges and (static or render or trained or segment or full_size or train_dual)" > gpurun_out/r5c8_parity_d1.txt 2>&1
tail -n 8 gpurun_out/r5c8_parity_d1.txt
