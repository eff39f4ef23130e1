#!/bin/bash
# round 5, call 13: x and x_res in the scaled ELU domain (three-instruction ELUs for all 512 of a row) against the current default
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python tools/abbench.py --frame --iters 20 --rounds 2 base x1 > gpurun_out/r5c13_ab.txt 2>&1
cat gpurun_out/r5c13_ab.txt
DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_x1.so timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not edges and not train" > gpurun_out/r5c13_parity.txt 2>&1; tail -6 gpurun_out/r5c13_parity.txt
