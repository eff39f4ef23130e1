#!/bin/bash
# round 6, call 4: the frame is not reproducible with two chunk streams (ragged or not; bit-reproducible on one): which kernel's output changes under a concurrent chunk?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/concurrency_probe.py 8192 > gpurun_out/r6c4_probe.txt 2>&1; tail -n 8 gpurun_out/r6c4_probe.txt | cut -c1-600
DYN_RAGGED=0 timeout 600 python tools/concurrency_probe.py 8192 > gpurun_out/r6c4_probe_noragged.txt 2>&1; tail -n 5 gpurun_out/r6c4_probe_noragged.txt | cut -c1-600
timeout 900 python tools/abbench.py --iters 20 --rounds 2 base e4 > gpurun_out/r6c4_ab.txt 2>&1; tail -n 4 gpurun_out/r6c4_ab.txt
