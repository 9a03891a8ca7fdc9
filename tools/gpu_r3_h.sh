#!/bin/bash
mkdir -p gpurun_out/r3h
O=$PWD/gpurun_out/r3h
timeout 600 python -m pytest tests/test_bigmlp_gpu.py -x -q -m gpu 2>&1 | tail -3
PQN_BM_STAMPS=1 python tools/bigmlp_gemm_bench.py 2048 1024 1024 128 2 20 2>&1 | tail -10
R=$PWD; cd /tmp && export TMPDIR=/tmp
for cfg in "2048 1024 1024 128 2" "2048 1024 1024 128 4" "1024 1024 1024 128 4" "1024 1024 1024 64 3" "2048 1024 1376 128 2" "1345 1024 1024 128 3"; do
  rm -rf /tmp/pg
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pg -o x -- python $R/tools/bigmlp_gemm_bench.py $cfg > /tmp/pg.log 2>&1
  echo "$cfg: $(python $R/tools/rocprof_summary.py /tmp/pg/x_results.db 8 2>/dev/null | grep bm_gemm | awk '{print $3, "us avg"}') $(grep 'max err' /tmp/pg.log)"
done | tee $O/gemm_bench.txt
cd $R; timeout 300 python tools/craftax_c5_run.py 800 2>&1 | tail -1 | cut -c1-150
