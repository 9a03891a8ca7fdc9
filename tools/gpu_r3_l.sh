#!/bin/bash
# Round 3 artefacts: the bench line, the kernel-trace summary of the same command, PMC traffic of the headline launch shape.
mkdir -p gpurun_out/r3l
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python bench.py > gpurun_out/r3l/bench.json 2> gpurun_out/r3l/bench.err
tail -c 600 gpurun_out/r3l/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb -o x -- python $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $R/gpurun_out/r3l/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocprof_summary.py /tmp/pb/x_results.db 16 > $R/gpurun_out/r3l/kernel_stats_seeds16_bf16x3.txt
cd $R
bash tools/pmc_bench.sh bf16x3 > gpurun_out/r3l/pmc_x3_seeds16.txt 2>&1
cp gpurun_out/pmc_train_kernel_bf16x3_seeds16.json gpurun_out/r3l/ 2>/dev/null
head -12 gpurun_out/r3l/kernel_stats_seeds16_bf16x3.txt | cut -c1-150; tail -2 gpurun_out/r3l/pmc_x3_seeds16.txt | cut -c1-1500
