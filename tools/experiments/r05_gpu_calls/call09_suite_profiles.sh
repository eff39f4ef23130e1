#!/bin/bash
# round 5, call 9: the GPU suite on the final build, K1 at 11 views inside the pipeline with P = 16 against the rule's P = 64, and the round's profile collection
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5c9_gpu_tests.txt 2>&1
tail -n 14 gpurun_out/r5c9_gpu_tests.txt
for P in "" 16 32; do
  DYN_PG_P=$P timeout 300 python bench.py --views 11 --steps 20 --warmup 3 --no-extra --no-traffic --cpu-rays 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V=11 DYN_PG_P=$P', d['ms_per_step'], d['kernels_avg_ms'])" >> gpurun_out/r5c9_k1_v11.txt
done
cat gpurun_out/r5c9_k1_v11.txt
timeout 2400 bash tools/collect_profiles.sh r05 > gpurun_out/r5c9_collect.log 2>&1
tail -n 5 gpurun_out/r5c9_collect.log
