#!/bin/bash
# round 6, call 16: ragged rows with the parallel segment planner and the unit header prefetched in the blend, against regular dense rows (alternating processes); plan kernels under the tracer
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "static_net or full_size" > gpurun_out/r6c16_parity.txt 2>&1; grep -v "of limit" gpurun_out/r6c16_parity.txt | tail -n 2 | cut -c1-200
timeout 1200 python tools/abbench.py --frame --iters 10 --rounds 3 base noragged=@DYN_RAGGED=0 > gpurun_out/r6c16_ab.txt 2>&1; tail -n 10 gpurun_out/r6c16_ab.txt | cut -c1-360
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6c16_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-rays 0 --no-extra --no-traffic --no-x6 --views 11 > $GRAFT_REPO_ROOT/gpurun_out/r6c16_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py stats $(find gpurun_out/r6c16_trace -name '*.db' | head -1) > gpurun_out/r6c16_v11_kernel_stats.txt 2>&1; head -n 13 gpurun_out/r6c16_v11_kernel_stats.txt | cut -c1-150
find gpurun_out/r6c16_trace -name '*.db' -delete
