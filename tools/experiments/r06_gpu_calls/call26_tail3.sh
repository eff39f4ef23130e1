#!/bin/bash
# round 6, call 26: the frame whose last chunk has exactly 3 rays, against the reference's frame
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "render_single_image_nvi or chunk_streams" > gpurun_out/r6c26_tail3.txt 2>&1; grep -v "of limit" gpurun_out/r6c26_tail3.txt | tail -n 8 | cut -c1-300
