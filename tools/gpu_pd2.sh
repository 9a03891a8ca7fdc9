#!/bin/bash
# opt-in paired-dgrad backward of the T1 pair kernel (PQN_T1_PD2=1): determinism + agreement with the f32-MFMA mode at
# 2 / 8 / 256 pairs, phase stamps, and the headline bench with and without it (one call, one box)
PQN_T1_PAIR=2 PQN_T1_PD2=1 BRIEF=1 timeout 200 python tools/debug_x3_conv.py 2>&1 | grep "^C 4"
PQN_T1_PAIR=2 PQN_T1_PD2=1 PQN_MODE=2 PQN_T1_STAMPS=1 timeout 200 python tools/ablate_train.py 2>&1 | grep "WG0\|grad("
mkdir -p gpurun_out/pd2
for v in 1 0; do
  PQN_T1_PD2=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-extras > gpurun_out/pd2/bench_$v.json 2> gpurun_out/pd2/bench_$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/pd2/bench_$v.json").read().strip().splitlines()[-1])
print("PQN_T1_PD2=$v: value %.4g  ms/step %.2f  T1 us %.1f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
done
