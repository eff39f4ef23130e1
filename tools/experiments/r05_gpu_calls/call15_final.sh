#!/bin/bash
# round 5, call 15: the GPU suite on the final tree and the round's profile collection (tools/collect_profiles.sh r05)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5c15_gpu_tests.txt 2>&1
tail -n 8 gpurun_out/r5c15_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5c15_smoke.txt 2>&1; tail -n 2 gpurun_out/r5c15_smoke.txt
rm -rf gpurun_out/prof_r05
timeout 2400 bash tools/collect_profiles.sh r05 > gpurun_out/r5c15_collect.log 2>&1
tail -n 3 gpurun_out/r5c15_collect.log
