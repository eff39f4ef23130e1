#!/bin/bash
# Decomposition builds of the training GEMM's ring form (developer experiment, run on the GPU box through gpurun after building here):
#   TR_RX bit 1: no result stores, 2: no operand requests inside the loop, 4: no MFMAs, 8: no epilogue, 16: no fragment conversions.
#   Results are wrong by construction; only the times mean something.
#   usage: bash tools/gemm_decompose.sh build   (here)   |   bash tools/gemm_decompose.sh run   (GPU box)
set -u
R=$(cd $(dirname $0)/.. && pwd)
C=$R/dynibar_amd/csrc
VARIANTS="1 2 4 8 10 16 30"
if [ "${1:-run}" = build ]; then
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$R/include -DTR_RX=$v -c $C/dyn_train.hip -o /tmp/dt_rx$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/dyn_geometry.o $C/dyn_nets.o $C/dyn_encoder.o /tmp/dt_rx$v.o -o $C/libdynibar_hip_rx$v.so
  done
else
  cd $R/tools
  for v in "" $VARIANTS; do
    echo "variant ${v:-product}"
    DYNIBAR_HIP_LIB=$C/libdynibar_hip${v:+_rx$v}.so DYNIBAR_TRAIN_GEMM=${GD_MODE:-auto} python gemmbench.py 2>&1 | grep "^K" | head -2
  done
fi
