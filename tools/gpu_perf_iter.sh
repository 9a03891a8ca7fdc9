#!/bin/bash
# One perf iteration on the GPU box: kernel parity tests, per-phase cycle stamps of T1 (operand mode $1, default 2),
# and a short headline bench in that mode.  Output under gpurun_out/iter/.
MODE=${1:-2}
NAME=$([ "$MODE" = 2 ] && echo bf16x3 || ([ "$MODE" = 1 ] && echo f16 || echo f32))
mkdir -p gpurun_out/iter
timeout 600 python -m pytest tests/test_qnet_gpu.py -x -q -m gpu 2>&1 | tail -4
PQN_MODE=$MODE PQN_T1_STAMPS=1 timeout 300 python tools/ablate_train.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/iter/stamps_$NAME.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --matmul-dtype $NAME > gpurun_out/iter/bench_$NAME.json 2> gpurun_out/iter/bench_$NAME.err
python - <<PY
import json
d = json.loads(open("gpurun_out/iter/bench_$NAME.json").read().strip().splitlines()[-1])
print("value %.4g  ms/step %.2f  T1 us %.1f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
if [ "$2" = prof ]; then
  R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pi
  rocprofv3 --kernel-trace -d /tmp/pi -o x -- python $R/bench.py --steps 6 --warmup 2 --no-extras --matmul-dtype $NAME > /dev/null 2>&1
  python $R/tools/rocprof_summary.py /tmp/pi/x_results.db 8 | cut -c1-150 | tee $R/gpurun_out/iter/kernel_stats_$NAME.txt
fi
