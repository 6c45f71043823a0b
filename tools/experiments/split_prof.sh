#!/bin/bash
# Who waits for whom inside select_puct_split_kernel: libtamago_splitprof.so = the library with search.hip compiled -DTG_SPLIT_PROF
# (build here, run tools/experiments/split_prof.py on the GPU box)
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/experiments/_bin build/exp
OBJS=$(ls build/obj/*.o | grep -v "search.hip")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -DTG_SPLIT_PROF $SPLIT_PROF_EXTRA -x hip -c tamago_amd/csrc/search.hip -o build/exp/search_prof.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/experiments/_bin/libtamago_splitprof.so $OBJS build/exp/search_prof.o
ls -la tools/experiments/_bin/libtamago_splitprof.so
