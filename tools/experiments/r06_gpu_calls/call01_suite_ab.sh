#!/bin/bash
# round 6, call 1: the GPU suite on the cleaned sources + ragged dense rows + the product mask in the blend; same-box A/B of the round-5 library against
# this one and of the issue-priority forms (mode 5 = the exec-masked form of rounds 3-5, mode 1 = its clean equivalent, mode 0 = none); ragged rows on / off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6c1_gpu_tests.txt 2>&1
tail -n 15 gpurun_out/r6c1_gpu_tests.txt
timeout 1500 python tools/abbench.py --frame --iters 20 --rounds 2 r5 base p1 p0 > gpurun_out/r6c1_ab.txt 2>&1
tail -n 14 gpurun_out/r6c1_ab.txt
DYN_RAGGED=0 timeout 600 python tools/abbench.py --frame --iters 10 --rounds 1 base > gpurun_out/r6c1_ab_noragged.txt 2>&1
tail -n 4 gpurun_out/r6c1_ab_noragged.txt
