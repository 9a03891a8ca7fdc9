#!/bin/bash
mkdir -p gpurun_out/r2b
timeout 600 python tools/debug_e2e_groups.py 4096 32 32 > gpurun_out/r2b/e2e_4096.txt 2>&1
timeout 600 python tools/debug_e2e_groups.py 1024 32 32 > gpurun_out/r2b/e2e_1024.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --deselect "tests/test_parity_gpu.py::test_make_train_end_to_end_vs_oracle" > gpurun_out/r2b/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r2b/pytest.txt
tail -25 gpurun_out/r2b/pytest.txt; cat gpurun_out/r2b/e2e_4096.txt | head -40; head -5 gpurun_out/r2b/e2e_1024.txt
