#!/bin/bash
# round 5, call 16: the GPU suite with the chunk-stream invariance test, and the bench line of the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5c16_gpu_tests.txt 2>&1
tail -n 4 gpurun_out/r5c16_gpu_tests.txt
( time timeout 1200 python bench.py > gpurun_out/r5c16_bench.json 2> gpurun_out/r5c16_bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c16_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value',round(d['value']),'ms',round(d['ms_per_step'],4),'frac',round(r['frac'],4))
print('secondary',{k:round(v.get('frac'),4) for k,v in r['secondary'].items()})
print('state',r['state'].get('se_busy_fraction'), r['state']['any_slow'])
f=d['extra']['frame_nvi_288x512']
print('frame',round(f['ms_per_frame'],1),f.get('chunk_streams'),f['per_rank'].get('chunk_ms_rank0'))
PY
