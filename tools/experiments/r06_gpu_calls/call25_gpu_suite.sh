#!/bin/bash
# round 6, call 25: the whole GPU suite and the smoke with the cross-axis change in
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r6c25_gpu_suite.txt 2>&1; grep -v "of limit" gpurun_out/r6c25_gpu_suite.txt | tail -n 6 | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
