#!/bin/bash
# round 6, calls 40-42: the GPU suite twice in a row on one box (flakiness of the run-spread based gradient checks, box-to-box variance): one call per box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2; do
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/soak_$i.txt 2>&1
  grep -v "of limit" gpurun_out/soak_$i.txt | tail -n 1 | cut -c1-200
  grep "of limit" gpurun_out/soak_$i.txt | head -n 2 | cut -c1-160
done
