#!/bin/bash
cd purejaxql_amd/csrc
for v in "1 1" "1 0" "0 1"; do
  set -- $v
  rm -f pqn_qnet.o
  make HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -DPQN_DBG_P1=$1 -DPQN_DBG_P6=$2" > /dev/null 2>&1
  echo "== P1=$1 P6=$2"
  (cd ../..; timeout 300 python tools/debug_x3_conv.py 2>&1 | grep "mode 2")
done
