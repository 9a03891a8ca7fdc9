#!/bin/bash
mkdir -p gpurun_out/r2k
R=$GRAFT_REPO_ROOT
for md in f32 bf16x3; do PQN_MATMUL=$md timeout 300 python tools/suite_10seeds.py Breakout-MinAtar Asterix-MinAtar >> gpurun_out/r2k/learning.txt 2>&1; done
timeout 600 python tools/craftax_c5_run.py 3000 > gpurun_out/r2k/craftax_c5.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for md in bf16x3 f32; do
  rm -rf /tmp/pb
  rocprofv3 --kernel-trace -d /tmp/pb -o x -- python $R/bench.py --steps 20 --warmup 5 --no-extras --matmul-dtype $md > $R/gpurun_out/r2k/bench_prof_$md.json 2>/dev/null
  python $R/tools/rocprof_summary.py /tmp/pb/x_results.db 16 > $R/gpurun_out/r2k/kernel_stats_seeds16_$md.txt
done
rm -rf /tmp/pc
rocprofv3 --kernel-trace -d /tmp/pc -o x -- python $R/tools/craftax_c5_run.py 400 > /dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/pc/x_results.db 14 > $R/gpurun_out/r2k/kernel_stats_craftax_c5.txt
cd $R
PQN_MODE=2 bash tools/pmc_train.sh > gpurun_out/r2k/pmc_x3.txt 2>&1
PQN_MODE=0 bash tools/pmc_train.sh > gpurun_out/r2k/pmc_f32.txt 2>&1
cp gpurun_out/pmc_train_kernel_*.json gpurun_out/r2k/ 2>/dev/null
cat gpurun_out/r2k/learning.txt; cat gpurun_out/r2k/craftax_c5.txt | tail -2; head -12 gpurun_out/r2k/kernel_stats_seeds16_bf16x3.txt | cut -c1-150; head -12 gpurun_out/r2k/kernel_stats_craftax_c5.txt | cut -c1-150; tail -3 gpurun_out/r2k/pmc_x3.txt | cut -c1-600
