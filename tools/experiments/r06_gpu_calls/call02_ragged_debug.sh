#!/bin/bash
# round 6, call 2: the full-frame oracle check failed at one ray of 262 with ragged rows (1.2 of its limit): ragged against regular dense rows on the same inputs at
# frame-chunk scale; the failing test with DYN_RAGGED=0; A/B of the blend prefetch (e3) and of the earlier request of the gathered features (e2)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ragged_ab.py 8192 64 11 > gpurun_out/r6c2_ragged_ab.txt 2>&1; tail -n 30 gpurun_out/r6c2_ragged_ab.txt
timeout 600 python tools/ragged_ab.py 1024 128 11 > gpurun_out/r6c2_ragged_ab_s128.txt 2>&1; tail -n 12 gpurun_out/r6c2_ragged_ab_s128.txt
DYN_RAGGED=0 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_frames" > gpurun_out/r6c2_fullframe_noragged.txt 2>&1; tail -n 5 gpurun_out/r6c2_fullframe_noragged.txt
timeout 1500 python tools/abbench.py --frame --iters 20 --rounds 2 base e3 e2 > gpurun_out/r6c2_ab.txt 2>&1; tail -n 10 gpurun_out/r6c2_ab.txt
for t in e3 e2; do DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_$t.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "static_net or static_pass" > gpurun_out/r6c2_parity_$t.txt 2>&1; tail -n 3 gpurun_out/r6c2_parity_$t.txt; done
