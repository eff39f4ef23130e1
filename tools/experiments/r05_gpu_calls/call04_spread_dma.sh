#!/bin/bash
# round 5, call 4: the two-slot ring's LDS-DMA spread over the chunk (B6_DMA_SPREAD) against the burst behind the barrier; window 2/3, 1/2, whole chunk;
# parity of the spread build; the composite backward with direct suffix sums (tools/grad_rootcause.py)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/abbench.py --frame --iters 20 --rounds 2 r4 s0 s1 s1p0 s1h s1f > gpurun_out/r5c4_ab.txt 2>&1
tail -n 8 gpurun_out/r5c4_ab.txt
export DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_s1.so
timeout 600 python tools/grad_rootcause.py > gpurun_out/r5c4_grad.txt 2>&1
sed -n '/^(1)/,/^(2)/p' gpurun_out/r5c4_grad.txt | head -12
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not edges and (mlp_engine or static_pass or static_net or trained_scale or dynamic_net or render_rays_mv or train_dual or train_static_step or train_composite or segment_widths)" > gpurun_out/r5c4_parity_s1.txt 2>&1
tail -n 25 gpurun_out/r5c4_parity_s1.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "edges" > gpurun_out/r5c4_parity_edges.txt 2>&1
tail -n 30 gpurun_out/r5c4_parity_edges.txt
