#!/bin/bash
mkdir -p gpurun_out/r3q
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for extra in "" "_FUSED_OPT=True" "MATMUL_DTYPE=bf16x3"; do
  echo "== $extra"; timeout 300 python tools/time_default_run.py 1 1 0 $extra 2>&1 | tail -1 | tee -a gpurun_out/r3q/default_variants.txt
done
