#!/bin/bash
# round 6, call 23: the reference's torch.cross axis reproduced in the kernels -- the new tests, the whole GPU suite, and prev (HEAD before the change) against base
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cross_axis or train_static_step" > gpurun_out/r6c23_cross.txt 2>&1; grep -v "of limit" gpurun_out/r6c23_cross.txt | tail -n 3 | cut -c1-300
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r6c23_gpu_suite.txt 2>&1; grep -v "of limit" gpurun_out/r6c23_gpu_suite.txt | tail -n 3 | cut -c1-300
timeout 900 python tools/abbench.py --frame --iters 20 --rounds 2 prev base > gpurun_out/r6c23_ab.txt 2>&1; tail -n 4 gpurun_out/r6c23_ab.txt | cut -c1-200
