#!/bin/bash
# round 4, call B: fc1 weight gradient without split-K partials (qnet_fc1_wgrad_x3_kernel<true>) + dz handed over as
# pre-split bf16 planes: parity, A/B against the partial-slab form (PQN_T2_ACC=0), kernel trace
O=gpurun_out/r4b; mkdir -p $O
timeout 1200 python -m pytest tests/test_headline_gpu.py tests/test_qnet_gpu.py -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
run() {
  name=$1; shift
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name: value %.4g  ms/step %.2f  T1 us %.1f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"]))
except Exception as e:
    print("$name: FAILED", repr(e)); print(open("$O/bench_$name.err").read()[-1500:])
PY
}
PQN_T2_ACC=0 run acc0
PQN_T2_ACC=1 run acc1
PQN_T2_ACC=0 run acc0b
PQN_T2_ACC=1 run acc1b
R=$PWD
for v in 0 1; do
(cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pv; PQN_T2_ACC=$v timeout 600 rocprofv3 --kernel-trace -d /tmp/pv -o x -- python $R/bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1; python $R/tools/rocprof_summary.py /tmp/pv/x_results.db 8 | cut -c1-150) > $O/kernel_stats_acc$v.txt 2>&1
tail -9 $O/kernel_stats_acc$v.txt
done
