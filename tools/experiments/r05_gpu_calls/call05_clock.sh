#!/bin/bash
# round 5, call 5: is the 19 % of "no weight DMA" cycles or clock?  GRBM_GUI_ACTIVE (cycles) and the duration of k_static_views with the ring (L0) and with
# the barrier but no DMA (e5); plus legacy priority + spread DMA (L1) and the edge-of-range tests as restated
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
for T in L0 e5; do
  D=/tmp/pmc_$T; rm -rf $D; mkdir -p $D
  (cd $D && DYNIBAR_HIP_LIB=$R/dynibar_amd/csrc/libdynibar_hip_$T.so timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $D -- python $R/bench.py --child --steps 6 --warmup 2 > $D/log.txt 2>&1)
  echo "== $T" >> gpurun_out/r5c5_clock.txt
  python tools/rocpd_summary.py pmc $(find $D -name '*.db' | head -1) 2>&1 | grep -E "k_static_views|k_net_points|k_static_blend" >> gpurun_out/r5c5_clock.txt
done
cat gpurun_out/r5c5_clock.txt
timeout 900 python tools/abbench.py --frame --iters 20 --rounds 2 r4 L0 L1 s1 > gpurun_out/r5c5_ab.txt 2>&1
tail -n 6 gpurun_out/r5c5_ab.txt
DYNIBAR_HIP_LIB=$R/dynibar_amd/csrc/libdynibar_hip_L1.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "edges" > gpurun_out/r5c5_parity_edges.txt 2>&1
tail -n 12 gpurun_out/r5c5_parity_edges.txt
