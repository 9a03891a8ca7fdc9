#!/bin/bash
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_craftax_gpu.py -q > gpurun_out/r2c/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c/pytest.txt
timeout 600 python tools/debug_e2e_groups.py 4096 32 32 > gpurun_out/r2c/e2e_4096.txt 2>&1
tail -30 gpurun_out/r2c/pytest.txt; tail -5 gpurun_out/r2c/e2e_4096.txt
