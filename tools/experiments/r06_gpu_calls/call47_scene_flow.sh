#!/bin/bash
# round 6, call 47: k_expected_scene_flow with its coefficients staged through LDS: the tests that hold exp_sf, then the frame's kernel table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r6c47_gpu_suite.txt 2>&1; grep -v "of limit" gpurun_out/r6c47_gpu_suite.txt | tail -n 2 | cut -c1-300
timeout 600 python tools/framebench.py --frames 3 2>&1 | grep -E "^frame|k_expected|k_render_flows|k_project_gather|total kernel" | cut -c1-160
