#!/bin/bash
# kernel tables of the launch shapes below the position-parallel form's threshold (yaml default, C2, one seed of 4096 envs)
R=$PWD
O=$R/gpurun_out/shapes; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in "128 1 100" "1024 1 60" "4096 1 30"; do
  set -- $s
  rm -rf /tmp/ps; timeout 300 rocprofv3 --kernel-trace -d /tmp/ps -o x -- python $R/tools/shape_run.py $1 $2 $3 > $O/run_$1_$2.txt 2>&1
  tail -1 $O/run_$1_$2.txt
  python $R/tools/rocprof_summary.py /tmp/ps/x_results.db 16 | cut -c1-200 > $O/kernel_stats_$1_$2.txt
  timeout 120 python $R/tools/shape_run.py $1 $2 $3 | tail -1
done
