#!/bin/bash
# round 6 call E: learning sanity at the bench shape with the package defaults (MATMUL_DTYPE auto -> bf16x3, own stream, evaluations ON,
# no host wait behind replays), then the launcher scripts end to end, then the remaining GPU test files
O=gpurun_out/r6e; mkdir -p $O
{ timeout 600 python tools/learn_headline.py Breakout-MinAtar 1e8 2>&1 | grep -v amdgpu.ids | tail -2
  timeout 300 python tools/learn_headline.py Asterix-MinAtar 2e7 2>&1 | grep -v amdgpu.ids | tail -2; } > $O/learning.txt 2>&1
cat $O/learning.txt
bash tools/e2e_smoke.sh > $O/e2e.txt 2>&1; tail -30 $O/e2e.txt
timeout 3000 python -m pytest tests/ -q -m gpu > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log; tail -8 $O/pytest_all.log
