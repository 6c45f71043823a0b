#!/bin/bash
# Ablation builds of the 19x19 pair kernel: libtamago_exp<N>.so = the library with net_forward_w1dband.hip compiled -DWB_ABL=N
# (what a class of riders costs: results are wrong, only the timing means something).  Run tools/experiments/wb_time.py <N> on the GPU box.
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/experiments/_bin build/exp
OBJS=$(ls build/obj/*.o | grep -v net_forward_w1dband)
for n in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 \
      -DWB_ABL=$n -x hip -c tamago_amd/csrc/net_forward_w1dband.hip -o build/exp/wb_abl$n.o 2>/dev/null &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/experiments/_bin/libtamago_exp$n.so $OBJS build/exp/wb_abl$n.o ) &
done
wait
ls -la tools/experiments/_bin/
