#!/bin/bash
# Phase stamps of the training-step kernels: libtamago_trainprof.so = the library with train.hip compiled -DTG_TRAIN_PROF
# (build here, run tools/experiments/train_prof.py on the GPU box)
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/experiments/_bin build/exp
OBJS=$(ls build/obj/*.o | grep -v "train.hip")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -DTG_TRAIN_PROF $TRAIN_PROF_EXTRA -x hip -c tamago_amd/csrc/train.hip -o build/exp/train_prof.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/experiments/_bin/libtamago_trainprof.so $OBJS build/exp/train_prof.o
ls -la tools/experiments/_bin/libtamago_trainprof.so
