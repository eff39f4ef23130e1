#!/bin/bash
# round 5, call 11: K1 with P = 8 against the new default P = 16, and the number of workgroups that share the streaming of the maps (512 / 2048)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for V in 8 11; do
for CFG in ":512" "8:512" "16:2048" "8:2048" "16:256"; do
  P=${CFG%%:*}; N=${CFG##*:}
  DYN_PG_P=$P DYN_PG_PREF=$N timeout 300 python bench.py --views $V --steps 20 --warmup 3 --no-extra --no-traffic --cpu-rays 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V=$V P=$P pref=$N step', round(d['ms_per_step'],4), 'k_project_gather us', round(d['kernels_avg_ms']['k_project_gather']*1e3,1))" >> gpurun_out/r5c11_k1.txt
done; done
cat gpurun_out/r5c11_k1.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "project_gather or static_pass or render_rays_mv or reference_matrices or projector" > gpurun_out/r5c11_parity.txt 2>&1; tail -3 gpurun_out/r5c11_parity.txt
DYN_PG_P=8 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "project_gather or static_pass or reference_matrices" > gpurun_out/r5c11_parity_p8.txt 2>&1; tail -3 gpurun_out/r5c11_parity_p8.txt
