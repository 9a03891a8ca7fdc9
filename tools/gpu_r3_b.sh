#!/bin/bash
# round 3, job B: wide-MLP kernels (csrc/pqn_bigmlp.hip) vs the oracle, the Craftax tests on the new backend, C5 timing
mkdir -p gpurun_out/r3b
O=gpurun_out/r3b
timeout 900 python -m pytest tests/test_bigmlp_gpu.py -x -q -m gpu > $O/pytest_bigmlp.log 2>&1
tail -25 $O/pytest_bigmlp.log
timeout 900 python -m pytest tests/test_craftax_gpu.py -q -m gpu > $O/pytest_craftax.log 2>&1
tail -15 $O/pytest_craftax.log
timeout 300 python tools/craftax_c5_run.py 1500 > $O/c5.log 2>&1
tail -3 $O/c5.log
