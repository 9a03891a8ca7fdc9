#!/bin/bash
# Round 3, call t: three-launch K-split form (4 positions per workgroup) against the single-tile kernel at 16 seeds x 8 tiles
mkdir -p gpurun_out/r3t
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for S in 16; do
  for ks in 0 2; do
    echo "== S=$S PQN_T1_KSPLIT=$ks"; PQN_T1_KSPLIT_TILES=100000 PQN_T1_KSPLIT=$ks timeout 600 python tools/time_default_run.py $S 1 0 2>&1 | tail -1 | cut -c1-90 | tee -a gpurun_out/r3t/default_16seeds_forms.txt
  done
done
