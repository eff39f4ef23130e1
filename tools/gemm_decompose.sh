#!/bin/bash
# Decomposition builds of the training GEMM (developer experiment, run on the GPU box through gpurun after building here):
#   TR_EXP bit 1: no MFMAs, 2: no global loads inside the k loop, 4: no result stores.  Results are wrong by construction; only the
#   times mean something.   usage: bash tools/gemm_decompose.sh build   (here)   |   bash tools/gemm_decompose.sh run   (GPU box)
set -u
R=$(cd $(dirname $0)/.. && pwd)
C=$R/dynibar_amd/csrc
if [ "${1:-run}" = build ]; then
  for v in 1 2 4 6 7; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DTR_EXP=$v -c $C/dyn_train.hip -o /tmp/dt_exp$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/dyn_geometry.o $C/dyn_nets.o $C/dyn_encoder.o /tmp/dt_exp$v.o -o $C/libdynibar_hip_exp$v.so
  done
else
  cd $R/tools
  for v in "" _exp1 _exp2 _exp4 _exp6 _exp7; do
    echo "variant ${v:-product}"
    DYNIBAR_HIP_LIB=$C/libdynibar_hip$v.so python gemmbench.py 2>&1 | grep "^K" | head -2
  done
fi
