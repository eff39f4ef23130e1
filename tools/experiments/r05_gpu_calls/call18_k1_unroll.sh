#!/bin/bash
# round 5, call 18: K1 at P = 16 inside the pipeline: tap unroll 2 / 4 / 8, one barrier vs three
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for V in 8 11; do
for T in base gu1 gu2 gu3 base gu2; do
  L=$PWD/dynibar_amd/csrc/libdynibar_hip_$T.so; [ $T = base ] && L=$PWD/dynibar_amd/csrc/libdynibar_hip.so
  DYNIBAR_HIP_LIB=$L timeout 300 python bench.py --views $V --steps 20 --warmup 3 --no-extra --no-traffic --cpu-rays 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V=$V $T step', round(d['ms_per_step'],4), 'k_project_gather us', round(d['kernels_avg_ms']['k_project_gather']*1e3,1))" >> gpurun_out/r5c19_k1.txt
done; done
cat gpurun_out/r5c19_k1.txt
