#!/bin/bash
# round 6 call D': in-kernel fold of the fc1 gradient rows (pos_fold) -- tests, then the bench with pos_fold = 1 / 0
O=gpurun_out/r6d; mkdir -p $O
timeout 1800 python -m pytest tests/test_headline_gpu.py tests/test_qnet_gpu.py tests/test_fullsize_gpu.py tests/test_run_gpu.py tests/test_parity_gpu.py -q -m gpu -x > $O/pytest2.log 2>&1; echo "rc=$?" >> $O/pytest2.log; tail -6 $O/pytest2.log
for f in 1 0 1; do
PQN_POS_FOLD=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_fold$f.json 2> $O/bench.err; F=$f python - <<'PY'
import json, os
f = os.environ["F"]
d = json.loads(open(f"gpurun_out/r6d/bench_fold{f}.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("pos_fold=%s value %.4g ms/update %.2f bwd %.1f fwd %.1f fwd+bwd %.1f" % (f, d["value"], d["ms_per_step"], r["avg_launch_us"], r.get("forward_kernel_us", 0), r.get("gather_forward_backward_us", 0)))
PY
done
(R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pv; timeout 600 rocprofv3 --kernel-trace -d /tmp/pv -o x -- python $R/bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1; python $R/tools/rocprof_summary.py /tmp/pv/x_results.db 8 | cut -c1-150 > $R/$O/kstats2.txt; cat $R/$O/kstats2.txt)
