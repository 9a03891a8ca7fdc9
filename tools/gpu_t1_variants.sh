#!/bin/bash
cd purejaxql_amd/csrc
for v in "-DT1_NO_H1T" "-DT1_NO_WLOAD" "-DT1_NO_H1T -DT1_NO_WLOAD"; do
  rm -f pqn_qnet.o
  make HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-variable $v" > /dev/null 2>&1
  echo "== variant '$v'"
  (cd ../..; PQN_MODE=2 PQN_T1_STAMPS=1 timeout 300 python tools/ablate_train.py 2>&1 | grep "WG0\|grad(")
done
