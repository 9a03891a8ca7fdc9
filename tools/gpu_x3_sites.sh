#!/bin/bash
cd purejaxql_amd/csrc
for v in 1 2; do
  rm -f pqn_qnet.o
  make HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-variable -DX3_VARIANT=0 -DX3_SITE=$v" > /dev/null 2>&1
  echo "== non-volatile only at site $v (1 = conv fwd, 2 = fc1)"
  (cd ../..; BRIEF=1 timeout 300 python tools/debug_x3_conv.py 2>&1 | grep "^C " | cut -c1-110)
done
