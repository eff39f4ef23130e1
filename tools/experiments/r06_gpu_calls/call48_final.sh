#!/bin/bash
# round 6, call 48: the GPU suite on the final tree, smoke, the round's profile collection (tools/collect_profiles.sh r06: rocprofv3 kernel stats + PMC passes of the bench command and of one frame,
# the frame breakdown, the full bench line)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6c48_gpu_tests.txt 2>&1
grep -v "of limit" gpurun_out/r6c48_gpu_tests.txt | tail -n 4 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6c48_smoke.txt 2>&1; tail -n 2 gpurun_out/r6c48_smoke.txt | cut -c1-400
rm -rf gpurun_out/prof_r06
timeout 2700 bash tools/collect_profiles.sh r06 > gpurun_out/r6c48_collect.log 2>&1
tail -n 3 gpurun_out/r6c48_collect.log
