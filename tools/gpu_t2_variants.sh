#!/bin/bash
cd purejaxql_amd/csrc
for v in "-DT2_NO_MFMA" "-DT2_NO_STAGE" "-DT2_NO_BARRIER" "-DT2_NO_STAGE -DT2_NO_BARRIER" "-DT2_NO_MFMA -DT2_NO_STAGE -DT2_NO_BARRIER"; do
  rm -f pqn_qnet.o
  make HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-variable $v" > /dev/null 2>&1
  echo "== variant '$v'"
  (cd ../..; timeout 300 python tools/t2_stamps.py 2>&1 | tail -1 | cut -c1-300)
done
