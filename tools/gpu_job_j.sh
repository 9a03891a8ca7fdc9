#!/bin/bash
mkdir -p gpurun_out/r2j
timeout 1200 python -m pytest tests/test_craftax_env_gpu.py -q -x > gpurun_out/r2j/pytest.txt 2>&1
echo "rc=$?" >> gpurun_out/r2j/pytest.txt
tail -30 gpurun_out/r2j/pytest.txt
