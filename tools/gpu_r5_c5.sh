#!/bin/bash
# round 5, C5: parity tests of the wide-MLP / Craftax path, then the C5 run with this build against library variants
# (purejaxql_amd/csrc/variants/libpqn_hip_<name>.so, e.g. the build before a change) alternating inside one call.
# usage: tools/gpu_r5_c5.sh [variant ...]
R=$PWD
O=$R/gpurun_out/r5c5; mkdir -p $O
L=purejaxql_amd/csrc/libpqn_hip.so
timeout 900 python -m pytest tests/test_bigmlp_gpu.py tests/test_craftax_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
cp $L /tmp/libpqn_default.so
run() { timeout 300 python tools/craftax_c5_run.py 2>&1 | grep "Craftax-Classic C5" | sed "s/^/$1: /" | cut -c1-190; }
{
  run default
  for v in "$@"; do cp purejaxql_amd/csrc/variants/libpqn_hip_$v.so $L; run $v; cp /tmp/libpqn_default.so $L; run default; done
} | tee $O/c5_ab.txt
(cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pc5; timeout 600 rocprofv3 --kernel-trace -d /tmp/pc5 -o x -- python $R/tools/craftax_c5_run.py > $O/c5_run.txt 2>&1; python $R/tools/rocprof_summary.py /tmp/pc5/x_results.db 14 | cut -c1-170) > $O/c5_kernel_stats.txt 2>&1
head -18 $O/c5_kernel_stats.txt
