#!/bin/bash
# round 6, call 7: what k_static_ref_feat writes when k_motion_mlp runs on another stream
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
DYN_RAGGED=0 timeout 900 python tools/concurrency_probe3.py > gpurun_out/r6c7_probe3.txt 2>&1; tail -n 60 gpurun_out/r6c7_probe3.txt | cut -c1-400
