#!/bin/bash
# Round 3, call t: where the K-split form stops paying: yaml-default MinAtar run with S seeds batched into the launches
mkdir -p gpurun_out/r3t
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for S in 2 4 6; do
  for ks in 1 0; do
    echo "== S=$S PQN_T1_KSPLIT=$ks"; PQN_T1_KSPLIT_TILES=100000 PQN_T1_KSPLIT=$ks timeout 600 python tools/time_default_run.py $S 1 0 2>&1 | tail -1 | cut -c1-90 | tee -a gpurun_out/r3t/default_seeds_crossover.txt
  done
done
