#!/bin/bash
mkdir -p gpurun_out/r2h
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_qnet_gpu.py -q -x -k "bf16x3" > gpurun_out/r2h/pytest_qnet.txt 2>&1
echo "rc=$?" >> gpurun_out/r2h/pytest_qnet.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -q -k "cartpole or CartPole or regression_pins or optimistic" > gpurun_out/r2h/pytest_cp.txt 2>&1
echo "rc=$?" >> gpurun_out/r2h/pytest_cp.txt
cd /tmp && export TMPDIR=/tmp
for m in 2; do
  rm -rf /tmp/pt
  PQN_MODE=$m rocprofv3 --kernel-trace -d /tmp/pt -o x -- python $R/tools/ablate_train.py > /dev/null 2>&1
  echo "== mode $m" >> $R/gpurun_out/r2h/prof.txt
  python $R/tools/rocprof_summary.py /tmp/pt/x_results.db 8 | grep -E "qnet|radam" | cut -c1-140 >> $R/gpurun_out/r2h/prof.txt
  PQN_MODE=$m PQN_T1_STAMPS=1 python $R/tools/ablate_train.py 2>&1 | grep -E "WG0|grad" >> $R/gpurun_out/r2h/prof.txt
done
cd $R
tail -6 gpurun_out/r2h/pytest_qnet.txt; tail -6 gpurun_out/r2h/pytest_cp.txt; cat gpurun_out/r2h/prof.txt
