#!/bin/bash
# round artefacts: kernel-trace summaries of the headline bench command (bf16x3 and f32 operand modes), PMC traffic of T1
mkdir -p gpurun_out/final
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for md in bf16x3 f32; do
  rm -rf /tmp/pb
  rocprofv3 --kernel-trace -d /tmp/pb -o x -- python $R/bench.py --steps 20 --warmup 5 --no-extras --matmul-dtype $md > $R/gpurun_out/final/bench_under_rocprof_$md.json 2>/dev/null
  python $R/tools/rocprof_summary.py /tmp/pb/x_results.db 16 > $R/gpurun_out/final/kernel_stats_seeds16_$md.txt
done
cd $R
bash tools/pmc_bench.sh bf16x3 > gpurun_out/final/pmc_x3_seeds16.txt 2>&1   # the headline launch shape (16 seeds per launch)
PQN_MODE=2 bash tools/pmc_train.sh > gpurun_out/final/pmc_x3.txt 2>&1
PQN_MODE=0 bash tools/pmc_train.sh > gpurun_out/final/pmc_f32.txt 2>&1
cp gpurun_out/pmc_train_kernel_*.json gpurun_out/final/ 2>/dev/null
PQN_MODE=2 PQN_T1_PAIR=2 PQN_T1_STAMPS=1 timeout 300 python tools/ablate_train.py 2>&1 | grep "WG\|grad(" > gpurun_out/final/t1_pair_stamps.txt
PQN_MODE=2 PQN_T1_PAIR=0 PQN_T1_STAMPS=1 timeout 300 python tools/ablate_train.py 2>&1 | grep "WG\|grad(" > gpurun_out/final/t1_single_stamps.txt
timeout 300 python tools/t2_stamps.py 2>&1 | tail -1 | cut -c1-600 > gpurun_out/final/t2_stamps.txt
head -12 gpurun_out/final/kernel_stats_seeds16_bf16x3.txt | cut -c1-150; tail -2 gpurun_out/final/pmc_x3_seeds16.txt | cut -c1-1200
