#!/bin/bash
# round 6, final artefacts: the bench line, the rocprofv3 kernel-trace summaries of the same command (16 seeds per launch) and of a
# one-seed run, the PMC passes of the position-parallel kernels (tools/pmc_pos.sh), the C5 kernel table.
# Copy gpurun_out/r6final/* to profiles/r06_*.
R=$PWD
O=$R/gpurun_out/r6final; mkdir -p $O
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
(cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pk; timeout 600 rocprofv3 --kernel-trace -d /tmp/pk -o x -- python $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > /dev/null 2>&1; python $R/tools/rocprof_summary.py /tmp/pk/x_results.db 20 | cut -c1-200) > $O/kernel_stats_seeds16_f16x2.txt 2>&1
head -12 $O/kernel_stats_seeds16_f16x2.txt | cut -c1-150
(cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pk1; timeout 600 rocprofv3 --kernel-trace -d /tmp/pk1 -o x -- python $R/bench.py --steps 20 --warmup 5 --seeds-per-gpu 1 --no-extras --no-cpu-baseline > $O/bench_seed1.json 2>/dev/null; python $R/tools/rocprof_summary.py /tmp/pk1/x_results.db 20 | cut -c1-200) > $O/kernel_stats_seed1_auto.txt 2>&1
head -8 $O/kernel_stats_seed1_auto.txt | cut -c1-150
(cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pkb; timeout 600 rocprofv3 --kernel-trace -d /tmp/pkb -o x -- python $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --matmul-dtype bf16x3 > /dev/null 2>&1; python $R/tools/rocprof_summary.py /tmp/pkb/x_results.db 8 | cut -c1-200) > $O/kernel_stats_seeds16_bf16x3.txt 2>&1
MD=f16x2 bash tools/pmc_pos.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_pos_bwd_kernel_f16x2_seeds16.json gpurun_out/pmc_pos_sq.txt $O/ 2>/dev/null; tail -c 1200 $O/pmc.log
(cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pc5; timeout 600 rocprofv3 --kernel-trace -d /tmp/pc5 -o x -- python $R/tools/craftax_c5_run.py > $O/c5_run.txt 2>&1; python $R/tools/rocprof_summary.py /tmp/pc5/x_results.db 24 | cut -c1-200) > $O/c5_kernel_stats.txt 2>&1
tail -3 $O/c5_run.txt
