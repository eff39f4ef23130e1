#!/bin/bash
# round 5, call 6: ray_dir_fc.0's per-point columns pooled (p5) and the weight pieces' cache policy (nt / sc1 / sc0 sc1) against legacy priority + spread DMA (L1)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/abbench.py --frame --iters 20 --rounds 2 r4 L1 p5 c_nt c_sc1 c_sc01 > gpurun_out/r5c6_ab.txt 2>&1
tail -n 8 gpurun_out/r5c6_ab.txt
DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_p5.so timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mlp_engine or static_pass or static_net or trained_scale or dynamic_net or render_rays_mv or segment_widths or full_size" > gpurun_out/r5c6_parity_p5.txt 2>&1
tail -n 12 gpurun_out/r5c6_parity_p5.txt
