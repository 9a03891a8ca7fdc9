#!/bin/bash
# A/B of library variants (tools/build_pos_variant.sh) under the headline bench inside ONE gpurun call (the boxes of the pool
# differ by +-4 %): value, ms per update, HIP-event durations of the backward / forward kernels.  usage: gpu_pos_ab.sh v1 v2 ...
O=gpurun_out/posab; mkdir -p $O
L=purejaxql_amd/csrc/libpqn_hip.so
cp $L /tmp/libpqn_default.so
run() {
  timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    r = d["roofline"]; t = r.get("training_step", {})
    print("%-12s value %.4g  ms/update %.2f  bwd %.1f us  fwd %.1f us  gather+fwd+bwd %.1f us  forms %s" % ("$1", d["value"], d["ms_per_step"], r["avg_launch_us"], t.get("forward_kernel_us", 0), t.get("gather_forward_backward_us", 0), d["config"].get("kernel_forms")))
except Exception as e:
    print("$1 failed", e); print(open("$O/bench_$1.err").read()[-1500:])
PY
}
run default
for v in "$@"; do
  cp purejaxql_amd/csrc/variants/libpqn_hip_$v.so $L
  run $v
done
cp /tmp/libpqn_default.so $L
run default2
