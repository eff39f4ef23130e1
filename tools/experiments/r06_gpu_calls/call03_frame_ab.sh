#!/bin/bash
# round 6, call 3: ragged == regular to 2e-7 on one 8192-ray call, yet the full-frame oracle check fails at one ray with ragged rows: whole frames under ragged on / off and one
# or two chunk streams, twice each, all rays compared; the failing test three times; the explicit priority pulse (p6) against the exec-masked form (base)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/ragged_frame_ab.py > gpurun_out/r6c3_frame_ab.txt 2>&1; tail -n 20 gpurun_out/r6c3_frame_ab.txt
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_frames" > gpurun_out/r6c3_fullframe_$i.txt 2>&1; grep -h "passed\|failed\|AssertionError: full" gpurun_out/r6c3_fullframe_$i.txt | tail -n 2; done
timeout 1200 python tools/abbench.py --frame --iters 20 --rounds 2 base p6 > gpurun_out/r6c3_ab.txt 2>&1; tail -n 4 gpurun_out/r6c3_ab.txt
