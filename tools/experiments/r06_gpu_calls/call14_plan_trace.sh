#!/bin/bash
# round 6, call 14: the plan kernels after the coalesced k_ragged_points (rocprofv3 kernel trace of the 11-view step), ragged on / off alternating on one box, ragged parity
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "static_net or full_frames or full_size" > gpurun_out/r6c14_parity.txt 2>&1; grep -v "of limit" gpurun_out/r6c14_parity.txt | tail -n 2 | cut -c1-200
timeout 1200 python tools/abbench.py --frame --iters 10 --rounds 3 base noragged=@DYN_RAGGED=0 > gpurun_out/r6c14_ab.txt 2>&1; tail -n 10 gpurun_out/r6c14_ab.txt | cut -c1-360
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6c14_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-rays 0 --no-extra --no-traffic --no-x6 --views 11 > $GRAFT_REPO_ROOT/gpurun_out/r6c14_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py stats $(find gpurun_out/r6c14_trace -name '*.db' | head -1) > gpurun_out/r6c14_v11_kernel_stats.txt 2>&1; head -n 16 gpurun_out/r6c14_v11_kernel_stats.txt | cut -c1-150
find gpurun_out/r6c14_trace -name '*.db' -delete
