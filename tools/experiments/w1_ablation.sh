#!/bin/bash
# Ablation builds of the 9x9 three-board kernel: libtamago_w1exp<N>.so = the library with net_forward_w1d.hip compiled -DW1_ABL=N
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/experiments/_bin build/exp
OBJS=$(ls build/obj/*.o | grep -v "net_forward_w1d.hip")
for n in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 \
      -DW1_ABL=$n -x hip -c tamago_amd/csrc/net_forward_w1d.hip -o build/exp/w1_abl$n.o 2>/dev/null &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/experiments/_bin/libtamago_w1exp$n.so $OBJS build/exp/w1_abl$n.o ) &
done
wait
ls tools/experiments/_bin/
