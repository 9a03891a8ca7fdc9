#!/bin/bash
# round 5, C5: the wide-MLP GEMM with ONE LDS stage (option bm_stages = 1: four 64 x 64 workgroups per CU) against the two-stage
# default -- parity tests under both, then the C5 run alternating between them inside one call.
R=$PWD
O=$R/gpurun_out/r5c5; mkdir -p $O
PQN_BM_STAGES=1 timeout 900 python -m pytest tests/test_bigmlp_gpu.py -x -q -m gpu > $O/tests_stages1.txt 2>&1; tail -3 $O/tests_stages1.txt
for st in 2 1 2 1; do
  PQN_BM_STAGES=$st timeout 300 python tools/craftax_c5_run.py 2>&1 | grep "Craftax-Classic C5" | sed "s/^/bm_stages=$st: /" | cut -c1-200
done | tee $O/c5_stages_ab.txt
