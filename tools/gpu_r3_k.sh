#!/bin/bash
# Round 3, call k: the whole-update enqueue of the Craftax script's loop (pqn_bigmlp_update) -- tests, C5 rate with and
# without it, rocprof kernel stats of the driver run.
mkdir -p gpurun_out/r3k
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_craftax_gpu.py tests/test_craftax_env_gpu.py tests/test_bigmlp_gpu.py tests/test_parity_gpu.py -q -x -k "craftax or optimistic or bigmlp or wide or c5" -s > gpurun_out/r3k/pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r3k/pytest.txt
tail -25 gpurun_out/r3k/pytest.txt
timeout 300 python tools/craftax_c5_run.py 2000 1 2>&1 | tail -2 | tee gpurun_out/r3k/c5_driver.txt
timeout 300 python tools/craftax_c5_run.py 2000 0 2>&1 | tail -2 | tee gpurun_out/r3k/c5_stepwise.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pc5 -o x -- python $GRAFT_REPO_ROOT/tools/craftax_c5_run.py 450 1 > /tmp/pc5.log 2>&1; tail -2 /tmp/pc5.log)
python tools/rocprof_summary.py /tmp/pc5/x_results.db 40 > gpurun_out/r3k/c5_kernel_stats.txt 2>&1 || ls -R /tmp/pc5 | head
head -40 gpurun_out/r3k/c5_kernel_stats.txt
