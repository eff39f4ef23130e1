#!/bin/bash
# round 6, call 13: ragged rows with the block-wise segment planner, fixed-trip reductions and the blend's table prefetch: parity of the network / frame tests, ragged on / off on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "static_net or static_pass or dynamic_net or render_rays or frame or stress or mono" > gpurun_out/r6c13_parity.txt 2>&1; grep -v "of limit" gpurun_out/r6c13_parity.txt | tail -n 4 | cut -c1-300
timeout 600 python tools/abbench.py --frame --iters 10 --rounds 2 base > gpurun_out/r6c13_ab_ragged.txt 2>&1; tail -n 3 gpurun_out/r6c13_ab_ragged.txt | cut -c1-360
DYN_RAGGED=0 timeout 600 python tools/abbench.py --frame --iters 10 --rounds 2 base > gpurun_out/r6c13_ab_noragged.txt 2>&1; tail -n 3 gpurun_out/r6c13_ab_noragged.txt | cut -c1-360
python - <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
from dynibar_amd import _lib
L = _lib.lib()
st = bench.StaticStep('cuda:0', 4096, 64, 11)
for _ in range(3): st.step()
torch.cuda.synchronize(); L.dyn_profile_enable(1)
for _ in range(10): st.step()
torch.cuda.synchronize(); L.dyn_profile_enable(0)
print('11 views, us per launch:', {k: round(v['avg_ms'] * 1e3, 1) for k, v in bench.read_kernels(L).items()})
PY
