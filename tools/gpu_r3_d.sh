#!/bin/bash
# C5 profile: per-kernel GPU time of the Craftax-Classic loop on the wide-MLP kernels
mkdir -p gpurun_out/r3d
O=$PWD/gpurun_out/r3d
timeout 300 python tools/craftax_c5_run.py 1500 > $O/c5.log 2>&1; tail -1 $O/c5.log
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pc5
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pc5 -o x -- python $R/tools/craftax_c5_run.py 400 > $O/prof_run.log 2>&1
tail -1 $O/prof_run.log
python $R/tools/rocprof_summary.py /tmp/pc5/x_results.db 40 > $O/c5_kernel_stats.txt 2>&1 || ls -R /tmp/pc5 | head
cut -c1-150 $O/c5_kernel_stats.txt | head -50
