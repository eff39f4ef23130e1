#!/bin/bash
# round 6, call 28: the static pass against the float64 oracle (kernels held to twice the fp32 oracle's own distance from it)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "accuracy_against_float64" > gpurun_out/r6c28_fp64.txt 2>&1; grep -E "accuracy against|passed|failed|Error" gpurun_out/r6c28_fp64.txt | cut -c1-600
