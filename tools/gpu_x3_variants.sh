#!/bin/bash
cd purejaxql_amd/csrc
for v in ${@:-0 1 2 3 4}; do
  rm -f pqn_qnet.o
  make HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-variable -DX3_VARIANT=$v" > /dev/null 2>&1
  echo "== X3_VARIANT=$v"
  (cd ../..; BRIEF=1 timeout 300 python tools/debug_x3_conv.py 2>&1 | grep "^C ")
done
