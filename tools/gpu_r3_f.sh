#!/bin/bash
# GEMM kernel time per shape / tile / split
mkdir -p gpurun_out/r3f
O=$PWD/gpurun_out/r3f
R=$PWD; cd /tmp && export TMPDIR=/tmp
for cfg in "2048 1024 1024 128 3" "2048 1024 1024 128 1" "2048 1024 1024 64 2" "1024 1024 1024 128 4" "1024 1024 1024 64 3" "1024 1024 1024 64 1" "2048 1024 1376 128 3" "1345 1024 1024 64 3"; do
  rm -rf /tmp/pg
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pg -o x -- python $R/tools/bigmlp_gemm_bench.py $cfg > /tmp/pg.log 2>&1
  echo "$cfg: $(python $R/tools/rocprof_summary.py /tmp/pg/x_results.db 8 2>/dev/null | grep bm_gemm | awk '{print $3, "us avg"}') $(grep 'max err' /tmp/pg.log)"
done | tee $O/gemm_bench.txt
