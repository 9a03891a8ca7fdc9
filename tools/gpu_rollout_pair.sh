#!/bin/bash
# Pair form of the rollout kernel: parity suites with the pair form forced on at every size it supports, then the
# headline bench with and without it.  Output under gpurun_out/rp/.
mkdir -p gpurun_out/rp
PQN_ROLLOUT_PAIR=2 timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_qnet_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/rp/tests_forced.txt
for v in 1 0; do
  PQN_ROLLOUT_PAIR=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-extras > gpurun_out/rp/bench_$v.json 2> gpurun_out/rp/bench_$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/rp/bench_$v.json").read().strip().splitlines()[-1])
print("PQN_ROLLOUT_PAIR=$v value %.4g  ms/step %.2f" % (d["value"], d["ms_per_step"]))
PY
done
