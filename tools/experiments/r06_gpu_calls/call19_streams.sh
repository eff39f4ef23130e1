#!/bin/bash
# round 6, call 19: one, two and three chunk streams with the final library (alternating processes)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/abbench.py --frame --iters 5 --rounds 3 base s1=@DYNIBAR_CHUNK_STREAMS=1 s3=@DYNIBAR_CHUNK_STREAMS=3 > gpurun_out/r6c19_ab_streams.txt 2>&1; tail -n 5 gpurun_out/r6c19_ab_streams.txt | cut -c1-200
