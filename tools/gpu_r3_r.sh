#!/bin/bash
# Round 3, call r: K-split form of the training kernel for small minibatches: parity tests, then the yaml-default run
mkdir -p gpurun_out/r3r
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_qnet_gpu.py -q -x -k "ksplit or grad_vs_oracle or bf16x3_is_det" > gpurun_out/r3r/pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r3r/pytest.txt
tail -25 gpurun_out/r3r/pytest.txt
for ks in 1 0; do
  echo "== PQN_T1_KSPLIT=$ks"; PQN_T1_KSPLIT=$ks timeout 300 python tools/time_default_run.py 1 1 0 2>&1 | tail -1 | tee -a gpurun_out/r3r/default_ksplit.txt
done
