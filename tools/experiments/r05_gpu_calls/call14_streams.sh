#!/bin/bash
# round 5, call 14: the chunks of a frame on 1 / 2 / 3 alternating HIP streams (render_image.CHUNK_STREAMS)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for N in 1 2 3 2 1; do
  echo "== DYNIBAR_CHUNK_STREAMS=$N" >> gpurun_out/r5c14_streams.txt
  DYNIBAR_CHUNK_STREAMS=$N timeout 600 python tools/framebench.py --frames 3 2>&1 | grep -E "^frame|total kernel" >> gpurun_out/r5c14_streams.txt
done
cat gpurun_out/r5c14_streams.txt
DYNIBAR_CHUNK_STREAMS=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q -m gpu -k "render_single_image or full_frames or full_size or dist or checkpoint" > gpurun_out/r5c14_parity.txt 2>&1; tail -4 gpurun_out/r5c14_parity.txt
