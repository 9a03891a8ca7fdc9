#!/bin/bash
# profile T1 / T2 per operand mode (kernel trace + phase stamps), then the changed tests
mkdir -p gpurun_out/r2f
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in 0 2; do
  rm -rf /tmp/pt
  PQN_MODE=$m rocprofv3 --kernel-trace -d /tmp/pt -o x -- python $R/tools/ablate_train.py > /dev/null 2>&1
  echo "== mode $m" >> $R/gpurun_out/r2f/prof.txt
  python $R/tools/rocprof_summary.py /tmp/pt/x_results.db 8 | grep -E "qnet|radam" | cut -c1-140 >> $R/gpurun_out/r2f/prof.txt
  PQN_MODE=$m PQN_T1_STAMPS=1 python $R/tools/ablate_train.py 2>&1 | grep -E "WG0|grad" >> $R/gpurun_out/r2f/prof.txt
done
cd $R
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_craftax_gpu.py -q -k "cartpole or CartPole or regression_pins or optimistic or trajectory or (end_to_end and extra8) or craftax" > gpurun_out/r2f/pytest.txt 2>&1
echo "rc=$?" >> gpurun_out/r2f/pytest.txt
cat gpurun_out/r2f/prof.txt; tail -12 gpurun_out/r2f/pytest.txt
