#!/bin/bash
# Round 3, call p: where the yaml-default MinAtar run (128 envs, 1e7 steps) spends its time
mkdir -p gpurun_out/r3p
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 300 python tools/time_default_run.py 1 1 1 2>&1 | tail -1 | tee gpurun_out/r3p/default_test1.txt
timeout 300 python tools/time_default_run.py 1 1 0 2>&1 | tail -1 | tee gpurun_out/r3p/default_test0.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pd -o x -- python $GRAFT_REPO_ROOT/tools/time_default_run.py 1 1 0 > /tmp/pd.log 2>&1; tail -1 /tmp/pd.log)
python tools/rocprof_summary.py /tmp/pd/x_results.db 24 > gpurun_out/r3p/default_kernel_stats.txt 2>&1
head -30 gpurun_out/r3p/default_kernel_stats.txt | cut -c1-150
