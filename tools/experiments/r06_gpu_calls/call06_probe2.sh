#!/bin/bash
# round 6, call 6: the static net's output changes under a concurrent chunk (rounds 5 and 6 alike): which of its kernels, under which concurrent kernel, dense (11) and lane-segment (8) flavour
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
DYN_RAGGED=0 timeout 900 python tools/concurrency_probe2.py 11 > gpurun_out/r6c6_probe2_v11.txt 2>&1; tail -n 24 gpurun_out/r6c6_probe2_v11.txt | cut -c1-700
DYN_RAGGED=0 timeout 900 python tools/concurrency_probe2.py 8 > gpurun_out/r6c6_probe2_v8.txt 2>&1; tail -n 24 gpurun_out/r6c6_probe2_v8.txt | cut -c1-700
