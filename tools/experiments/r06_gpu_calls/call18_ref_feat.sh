#!/bin/bash
# round 6, call 18: k_static_ref_feat with the embedding formed once per ray: parity, time, the co-runner x victim matrix and two-stream frames again
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "static_net or static_pass or full_size or full_frames or bench_shape" > gpurun_out/r6c18_parity.txt 2>&1; grep -v "of limit" gpurun_out/r6c18_parity.txt | tail -n 2 | cut -c1-200
DYN_RAGGED=0 timeout 600 python tools/concurrency_probe4.py > gpurun_out/r6c18_probe4.txt 2>&1; grep "co-runner\|alone" gpurun_out/r6c18_probe4.txt | cut -c1-330
timeout 600 python tools/ragged_frame_ab.py --quick > gpurun_out/r6c18_frame_ab.txt 2>&1; grep "vs\|same process" gpurun_out/r6c18_frame_ab.txt | cut -c1-220
timeout 600 python tools/abbench.py --iters 20 --rounds 2 base > gpurun_out/r6c18_ab.txt 2>&1; tail -n 4 gpurun_out/r6c18_ab.txt | cut -c1-300
