#!/bin/bash
# Round 3 final: whole -m gpu suite + smoke, then the round's artefacts (tools/gpu_r3_l.sh: bench line, kernel-trace
# summary of the bench command, PMC traffic of the headline launch shape), the C5 kernel stats and the default-run time.
mkdir -p gpurun_out/r3f
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r3f/pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r3f/pytest.txt
tail -14 gpurun_out/r3f/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3f/smoke.txt 2>&1
tail -2 gpurun_out/r3f/smoke.txt
bash tools/gpu_r3_l.sh > gpurun_out/r3f/artefacts.log 2>&1
tail -14 gpurun_out/r3f/artefacts.log | cut -c1-400
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pc5 -o x -- python $GRAFT_REPO_ROOT/tools/craftax_c5_run.py 450 1 > /tmp/pc5.log 2>&1; tail -1 /tmp/pc5.log | tee $GRAFT_REPO_ROOT/gpurun_out/r3f/c5_run.txt)
python tools/rocprof_summary.py /tmp/pc5/x_results.db 40 > gpurun_out/r3f/c5_kernel_stats.txt 2>&1
timeout 300 python tools/time_default_run.py 1 1 1 2>&1 | tail -1 | tee gpurun_out/r3f/default_test1.txt
