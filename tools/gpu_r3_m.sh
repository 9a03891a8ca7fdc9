#!/bin/bash
# Round 3, call m: backward of the wide MLP with its parameter-gradient side on a second stream + batched plane refresh.
mkdir -p gpurun_out/r3m
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bigmlp_gpu.py tests/test_craftax_gpu.py -q -x > gpurun_out/r3m/pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r3m/pytest.txt
tail -8 gpurun_out/r3m/pytest.txt
for ov in 1 0; do
  PQN_BM_OVERLAP=$ov timeout 300 python tools/craftax_c5_run.py 2000 1 2>&1 | tail -1 | tee gpurun_out/r3m/c5_overlap$ov.txt
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pc5 -o x -- python $GRAFT_REPO_ROOT/tools/craftax_c5_run.py 450 1 > /tmp/pc5.log 2>&1; tail -1 /tmp/pc5.log)
python tools/rocprof_summary.py /tmp/pc5/x_results.db 40 > gpurun_out/r3m/c5_kernel_stats.txt 2>&1
head -30 gpurun_out/r3m/c5_kernel_stats.txt | cut -c1-140
