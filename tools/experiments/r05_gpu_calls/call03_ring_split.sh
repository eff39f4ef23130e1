#!/bin/bash
# round 5, call 3: what of the ring costs the two-wave kernels 19 % -- the barrier (e5), the DMA + own-piece wait (e6) or the DMA alone (e7); no LDS A reads (e3);
# no MFMAs and no ring (e14); and the gradient root cause, extended
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/abbench.py --iters 20 --rounds 2 r4 e5 e6 e7 e3 e14 > gpurun_out/r5c3_ab.txt 2>&1
tail -n 8 gpurun_out/r5c3_ab.txt
timeout 600 python tools/grad_rootcause.py > gpurun_out/r5c3_grad.txt 2>&1
sed -n '/^(4)/,$p' gpurun_out/r5c3_grad.txt
