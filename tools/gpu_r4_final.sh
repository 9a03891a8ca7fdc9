#!/bin/bash
# round 4, final artefacts: the bench line, the rocprofv3 kernel-trace summary of the same command, PMC traffic of the headline
# launch shape (tools/pmc_bench.sh), the C5 kernel table.  Copy gpurun_out/r4final/* to profiles/r04_*.
R=$PWD
O=$R/gpurun_out/r4final; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
(cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pk; timeout 600 rocprofv3 --kernel-trace -d /tmp/pk -o x -- python $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > /dev/null 2>&1; python $R/tools/rocprof_summary.py /tmp/pk/x_results.db 20 | cut -c1-200) > $O/kernel_stats_seeds16_bf16x3.txt 2>&1
head -12 $O/kernel_stats_seeds16_bf16x3.txt | cut -c1-150
bash tools/pmc_bench.sh bf16x3 > $O/pmc.log 2>&1; cp gpurun_out/pmc_train_kernel_bf16x3_seeds16.json $O/ 2>/dev/null; tail -c 600 $O/pmc.log
(cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pc5; timeout 600 rocprofv3 --kernel-trace -d /tmp/pc5 -o x -- python $R/tools/craftax_c5_run.py > $O/c5_run.txt 2>&1; python $R/tools/rocprof_summary.py /tmp/pc5/x_results.db 24 | cut -c1-200) > $O/c5_kernel_stats.txt 2>&1
tail -3 $O/c5_run.txt
