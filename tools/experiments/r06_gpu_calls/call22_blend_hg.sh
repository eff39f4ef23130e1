#!/bin/bash
# round 6, call 22: the blend with the point part of rgb_fc.0 requested under the layer instead of in front of it (hg) against base; parity of the static net with the variant
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_hg.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "static_net or static_pass or bench_shape" > gpurun_out/r6c22_parity.txt 2>&1; grep -v "of limit" gpurun_out/r6c22_parity.txt | tail -n 2 | cut -c1-200
timeout 1200 python tools/abbench.py --frame --iters 20 --rounds 3 base hg > gpurun_out/r6c22_ab.txt 2>&1; tail -n 10 gpurun_out/r6c22_ab.txt | cut -c1-360
