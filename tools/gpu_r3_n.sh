#!/bin/bash
# Round 3, call n: the one-shot peer all-reduce (two processes on the one GPU) and the env-sharded update as one graph.
mkdir -p gpurun_out/r3n
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_dist_gpu.py -q -x -s > gpurun_out/r3n/pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r3n/pytest.txt
tail -40 gpurun_out/r3n/pytest.txt
