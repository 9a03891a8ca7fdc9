#!/bin/bash
mkdir -p gpurun_out/r3e
O=$PWD/gpurun_out/r3e
timeout 900 python -m pytest tests/test_bigmlp_gpu.py -x -q -m gpu > $O/pytest_bigmlp.log 2>&1; tail -8 $O/pytest_bigmlp.log
timeout 900 python -m pytest tests/test_craftax_gpu.py -q -m gpu > $O/pytest_craftax.log 2>&1; tail -4 $O/pytest_craftax.log
for v in "0 0" "64 0" "128 0" "64 1" "128 1" "0 2" "0 4"; do
  set -- $v
  echo "bm_tile=$1 bm_split=$2: $(PQN_BM_TILE=$1 PQN_BM_SPLIT=$2 timeout 300 python tools/craftax_c5_run.py 800 2>&1 | tail -1 | cut -c1-140)"
done
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pc5
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pc5 -o x -- python $R/tools/craftax_c5_run.py 400 > $O/prof_run.log 2>&1
python $R/tools/rocprof_summary.py /tmp/pc5/x_results.db 40 > $O/c5_kernel_stats.txt 2>&1
cut -c1-150 $O/c5_kernel_stats.txt | head -36
