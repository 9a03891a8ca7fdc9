#!/bin/bash
# round 6 call C: the tests added after the full-suite run, then the form-threshold sweep
O=gpurun_out/r6c; mkdir -p $O
timeout 1500 python -m pytest tests/test_headline_gpu.py -q -m gpu -s -k "whole_update_vs_oracle and Breakout" > $O/pytest_f64.log 2>&1; echo "rc=$?" >> $O/pytest_f64.log; grep -a "update vector vs\|passed\|failed\|rc=" $O/pytest_f64.log
timeout 1500 python -m pytest tests/test_dist_gpu.py -q -m gpu -k "falls_back or bench_gpus" > $O/pytest_dist.log 2>&1; echo "rc=$?" >> $O/pytest_dist.log; tail -4 $O/pytest_dist.log
timeout 600 python -m pytest tests/test_run_gpu.py tests/test_fullsize_gpu.py tests/test_replay_fault_gpu.py -q -m gpu > $O/pytest_misc.log 2>&1; echo "rc=$?" >> $O/pytest_misc.log; tail -4 $O/pytest_misc.log
timeout 900 python tools/form_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/form_sweep.txt
