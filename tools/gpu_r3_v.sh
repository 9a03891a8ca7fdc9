#!/bin/bash
# Round 3, call v: kernel tests after a change of the bf16x3 training kernels
mkdir -p gpurun_out/r3v
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_qnet_gpu.py tests/test_headline_gpu.py tests/test_fullsize_gpu.py -q > gpurun_out/r3v/pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r3v/pytest.txt
tail -6 gpurun_out/r3v/pytest.txt
