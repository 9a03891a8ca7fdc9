#!/bin/bash
# XCD-aware (seed, pair) mapping of the T1 pair kernel: headline bench with the mapping (default) and with the plain
# grid order (PQN_ABLATE_TRAIN=256), then the kernel tests.  Output under gpurun_out/xcd/.
mkdir -p gpurun_out/xcd
for v in 0 256 0 256; do
  PQN_ABLATE_TRAIN=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-extras > gpurun_out/xcd/bench_$v.json 2> gpurun_out/xcd/bench_$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/xcd/bench_$v.json").read().strip().splitlines()[-1])
print("PQN_ABLATE_TRAIN=$v value %.4g  ms/step %.2f  T1 us %.1f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
done
timeout 900 python -m pytest tests/test_qnet_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -4
