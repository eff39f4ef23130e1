#!/bin/bash
# round 6, call 31: pooling weights as expm1 differences -- the whole GPU suite (margins), then prev (exp) against base (expm1) on the step and the frame
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r6c31_gpu_suite.txt 2>&1; grep -v "of limit" gpurun_out/r6c31_gpu_suite.txt | tail -n 4 | cut -c1-300; grep "of limit" gpurun_out/r6c31_gpu_suite.txt | head -8 | cut -c1-200
timeout 900 python tools/abbench.py --frame --iters 20 --rounds 2 prev base > gpurun_out/r6c31_ab.txt 2>&1; tail -n 4 gpurun_out/r6c31_ab.txt | cut -c1-200
