#!/bin/bash
# round 6, call 45: compute_traj_pts fused into the gather and the flows: the new bit-exactness tests, the whole suite, a frame with and without (DYNIBAR_FUSED_TRAJ=0)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused" > gpurun_out/r6c45_fused.txt 2>&1; grep -v "of limit" gpurun_out/r6c45_fused.txt | tail -n 3 | cut -c1-300
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r6c45_gpu_suite.txt 2>&1; grep -v "of limit" gpurun_out/r6c45_gpu_suite.txt | tail -n 3 | cut -c1-300
for f in 1 0 1 0; do DYNIBAR_FUSED_TRAJ=$f timeout 600 python tools/framebench.py --frames 3 2>&1 | grep -E "^frame|k_trajectory|k_project_gather|k_render_flows|total kernel" | sed "s/^/fused=$f  /" ; done > gpurun_out/r6c45_frames.txt 2>&1; cat gpurun_out/r6c45_frames.txt | cut -c1-160
