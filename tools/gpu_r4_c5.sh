#!/bin/bash
# round 4, C5 (Craftax-Classic, 4 x 1024 network) launch-level work: tests of the wide-MLP / Craftax path, then the C5 run
# plain and under rocprofv3 --kernel-trace.  Artefact: profiles/r04_v7_c5_batched_backward.txt
R=$PWD
O=$R/gpurun_out/r4c5; mkdir -p $O
timeout 1200 python -m pytest tests/test_bigmlp_gpu.py tests/test_craftax_env_gpu.py tests/test_craftax_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 300 python tools/craftax_c5_run.py > $O/c5_plain.txt 2>&1; tail -2 $O/c5_plain.txt
PQN_UPD_OVERLAP=0 timeout 300 python tools/craftax_c5_run.py > $O/c5_plain_no_side_stream.txt 2>&1; tail -1 $O/c5_plain_no_side_stream.txt
(cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pc5; timeout 600 rocprofv3 --kernel-trace -d /tmp/pc5 -o x -- python $R/tools/craftax_c5_run.py > $O/c5_run.txt 2>&1; python $R/tools/rocprof_summary.py /tmp/pc5/x_results.db 32 | cut -c1-200) > $O/c5_kernel_stats.txt 2>&1
tail -3 $O/c5_run.txt; head -40 $O/c5_kernel_stats.txt | cut -c1-150
