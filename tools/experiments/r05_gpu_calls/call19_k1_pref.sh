#!/bin/bash
# round 5, call 19: K1 (P = 16, unroll 2) inside the pipeline: with / without the map prefetch, prefetch share 1024 / default / 4096; gather parity on the final geometry object
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for V in 8 11; do
for CFG in "default" "nopref" "1024" "4096" "default"; do
  unset DYN_PG_NOPREF DYN_PG_PREF
  [ $CFG = nopref ] && export DYN_PG_NOPREF=1
  [ $CFG = 1024 ] && export DYN_PG_PREF=1024
  [ $CFG = 4096 ] && export DYN_PG_PREF=4096
  timeout 300 python bench.py --views $V --steps 20 --warmup 3 --no-extra --no-traffic --cpu-rays 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V=$V $CFG step', round(d['ms_per_step'],4), 'k_project_gather us', round(d['kernels_avg_ms']['k_project_gather']*1e3,1))" >> gpurun_out/r5c20_k1.txt
done; done
unset DYN_PG_NOPREF DYN_PG_PREF
cat gpurun_out/r5c20_k1.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "project_gather or static_pass or render_rays_mv or reference_matrices or projector or full_size or train_static_step" > gpurun_out/r5c20_parity.txt 2>&1; tail -3 gpurun_out/r5c20_parity.txt
