#!/bin/bash
# Round 3, call s: K-split form -- position-group size A/B on the yaml-default run, parity tests, kernel stats
mkdir -p gpurun_out/r3s
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_qnet_gpu.py -q -x -k "ksplit or grad_vs_oracle" > gpurun_out/r3s/pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r3s/pytest.txt
tail -4 gpurun_out/r3s/pytest.txt
for ks in 1 4; do
  echo "== PQN_T1_KSPLIT=$ks"; PQN_T1_KSPLIT=$ks timeout 300 python tools/time_default_run.py 1 1 0 2>&1 | tail -1 | tee -a gpurun_out/r3s/default_ksplit.txt
done
(cd /tmp && PQN_T1_KSPLIT=4 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pd -o x -- python $GRAFT_REPO_ROOT/tools/time_default_run.py 1 1 0 > /tmp/pd.log 2>&1; tail -1 /tmp/pd.log)
python tools/rocprof_summary.py /tmp/pd/x_results.db 14 > gpurun_out/r3s/default_kernel_stats.txt 2>&1
head -18 gpurun_out/r3s/default_kernel_stats.txt | cut -c1-150
