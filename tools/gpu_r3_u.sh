#!/bin/bash
# Round 3, call u: Craftax tests after the functional-step change
mkdir -p gpurun_out/r3u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_craftax_env_gpu.py tests/test_craftax_gpu.py tests/test_parity_gpu.py -q -k "craftax or optimistic" > gpurun_out/r3u/pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r3u/pytest.txt
tail -6 gpurun_out/r3u/pytest.txt
