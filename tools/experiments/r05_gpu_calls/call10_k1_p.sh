#!/bin/bash
# round 5, call 10: K1 tile height inside the pipeline: 8 views (bench step) and the frame's two gathers (7 displaced + 11 static views) with P forced to 16 / 32 / 64
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for V in 8 7 15; do
for P in "" 16 32 64; do
  DYN_PG_P=$P timeout 300 python bench.py --views $V --steps 20 --warmup 3 --no-extra --no-traffic --cpu-rays 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V=$V DYN_PG_P=$P step', round(d['ms_per_step'],4), 'k_project_gather us', round(d['kernels_avg_ms']['k_project_gather']*1e3,1))" >> gpurun_out/r5c10_k1.txt
done; done
cat gpurun_out/r5c10_k1.txt
for P in "" 16 32; do
  echo "== frame DYN_PG_P=$P" >> gpurun_out/r5c10_k1_frame.txt
  DYN_PG_P=$P timeout 600 python tools/abbench.py --frame --iters 10 --rounds 1 base 2>&1 | grep "round 0" >> gpurun_out/r5c10_k1_frame.txt
done
cat gpurun_out/r5c10_k1_frame.txt
