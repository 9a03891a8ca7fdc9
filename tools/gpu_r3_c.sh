#!/bin/bash
mkdir -p gpurun_out/r3c
O=gpurun_out/r3c
timeout 600 python -m pytest tests/test_bigmlp_gpu.py -q -m gpu -k gemm > $O/pytest_gemm.log 2>&1
tail -12 $O/pytest_gemm.log
timeout 300 python tools/debug_bigmlp.py > $O/debug.log 2>&1
tail -25 $O/debug.log
timeout 300 python tools/debug_bigmlp.py 37 256 2 5 96 > $O/debug2.log 2>&1
tail -14 $O/debug2.log
