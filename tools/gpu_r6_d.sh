#!/bin/bash
# round 6 call D: once-per-epoch gather -- the tests that exercise the position-parallel form through the update drivers, then the bench
O=gpurun_out/r6d; mkdir -p $O
timeout 1800 python -m pytest tests/test_headline_gpu.py tests/test_qnet_gpu.py tests/test_fullsize_gpu.py tests/test_run_gpu.py tests/test_replay_fault_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6d/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g ms/update %.2f bwd %.1f fwd %.1f fwd+bwd(+gather) %.1f forms %s" % (d["value"], d["ms_per_step"], r["avg_launch_us"], r.get("forward_kernel_us", 0), r.get("gather_forward_backward_us", 0), d["config"]["kernel_forms"]))
PY
(R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pv; timeout 600 rocprofv3 --kernel-trace -d /tmp/pv -o x -- python $R/bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1; python $R/tools/rocprof_summary.py /tmp/pv/x_results.db 12 | cut -c1-150 > $R/$O/kstats.txt; cat $R/$O/kstats.txt)
