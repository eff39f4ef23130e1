#!/bin/bash
# round 5, call 7: the GPU suite on the new default build (pooled L1, spread DMA, composite backward fix, K1 with one barrier), the K1 tile sweep
# (one barrier vs three: base vs g0; P = 16 / 32 / 64), and the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5c7_gpu_tests.txt 2>&1
tail -n 30 gpurun_out/r5c7_gpu_tests.txt
timeout 900 python tools/k1sweep.py 8,11 ",16,32,64" base,g0 > gpurun_out/r5c7_k1.txt 2>&1
cat gpurun_out/r5c7_k1.txt
timeout 900 python tools/kbench.py > gpurun_out/r5c7_kbench.txt 2>&1; tail -n 20 gpurun_out/r5c7_kbench.txt
DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_g0.so timeout 900 python tools/kbench.py > gpurun_out/r5c7_kbench_g0.txt 2>&1; tail -n 20 gpurun_out/r5c7_kbench_g0.txt
timeout 1200 python bench.py > gpurun_out/r5c7_bench.json 2> gpurun_out/r5c7_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c7_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value',d['value'],'ms',d['ms_per_step'],'frac',r['frac'])
print('clock',r.get('clock'))
print('secondary',{k:(v.get('frac'),v.get('avg_launch_ms')) for k,v in r['secondary'].items()})
print('state',r['state'])
print('frame',d['extra']['frame_nvi_288x512'].get('ms_per_frame'))
print('cpu',d.get('cpu_baseline'))
PY
