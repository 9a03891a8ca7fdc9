#!/bin/bash
# Round 3, call t: K-split forms at many tiles (10 seeds x 8 tiles per launch), threshold lifted
mkdir -p gpurun_out/r3t
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for ks in 0 2 3 4; do
  echo "== S=10 PQN_T1_KSPLIT=$ks"; PQN_T1_KSPLIT_TILES=100000 PQN_T1_KSPLIT=$ks timeout 600 python tools/time_default_run.py 10 1 0 2>&1 | tail -1 | cut -c1-90 | tee -a gpurun_out/r3t/default_10seeds_forms.txt
done
