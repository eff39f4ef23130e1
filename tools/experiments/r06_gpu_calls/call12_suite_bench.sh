#!/bin/bash
# round 6, call 12: the GPU suite on the library with exclusive CUs for the one-wave-per-SIMD kernels (+ the full-size two-stream test), smoke, one full bench line,
# ragged rows on / off on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6c12_gpu_tests.txt 2>&1
grep -v "of limit" gpurun_out/r6c12_gpu_tests.txt | tail -n 12 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6c12_smoke.txt 2>&1; tail -n 2 gpurun_out/r6c12_smoke.txt | cut -c1-400
timeout 1500 python bench.py > gpurun_out/r6c12_bench.json 2> gpurun_out/r6c12_bench.err; python - <<'PY'
import json
try:
  d = json.loads(open('gpurun_out/r6c12_bench.json').read().strip().splitlines()[-1])
  r = d['roofline']
  print('value', d['value'], 'ms_per_step', d['ms_per_step'])
  print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (dict, list, str))})
  print('cpu', d.get('cpu_baseline', {}).get('value'), 'check', d.get('check_vs_oracle'), 'x6', d.get('x6_engine'))
  print('frame', (d['extra'].get('frame_nvi_288x512') or {}).get('ms_per_frame'), 'views_11', (d['extra'].get('views_11') or {}).get('ms_per_step'))
except Exception as e:
  print('bench parse failed', e); print(open('gpurun_out/r6c12_bench.err').read()[-1500:])
PY
DYN_RAGGED=0 timeout 600 python tools/abbench.py --frame --iters 10 --rounds 1 base > gpurun_out/r6c12_ab_noragged.txt 2>&1; tail -n 3 gpurun_out/r6c12_ab_noragged.txt | cut -c1-330
timeout 600 python tools/abbench.py --frame --iters 10 --rounds 1 base > gpurun_out/r6c12_ab_ragged.txt 2>&1; tail -n 3 gpurun_out/r6c12_ab_ragged.txt | cut -c1-330
