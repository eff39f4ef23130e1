#!/bin/bash
# round 6, call 20: ragged dense rows for the dynamic net at 5..7 views (up to 64 points per workgroup): parity, then against the 8-lane segments on one box (DYN_RAGGED=0 also switches
# the static net's ragged rows off: the static and dynamic view kernels are read separately)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dynamic_net or render_rays or frame or mono or stress or trained" > gpurun_out/r6c20_parity.txt 2>&1; grep -v "of limit" gpurun_out/r6c20_parity.txt | tail -n 3 | cut -c1-250
timeout 1200 python tools/abbench.py --frame --iters 5 --rounds 3 base noragged=@DYN_RAGGED=0 > gpurun_out/r6c20_ab.txt 2>&1; tail -n 10 gpurun_out/r6c20_ab.txt | cut -c1-360
