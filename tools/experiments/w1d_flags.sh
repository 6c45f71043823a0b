#!/bin/bash
# the 9x9 tower compiled with other scheduler strategies (round 6): forward-only rate of each build
cd $GRAFT_REPO_ROOT
echo -n "default: "; python tools/bench_net.py 9 65536 2>&1 | tail -1
for v in maxilp maxmem iterilp nounclust; do
  echo -n "$v: "; TAMAGO_HIP_LIB=$GRAFT_REPO_ROOT/tools/experiments/_bin/libtamago_w1d_$v.so python tools/bench_net.py 9 65536 2>&1 | tail -1
done
