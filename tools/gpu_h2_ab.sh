#!/bin/bash
# in-call A/B of the two operand modes of the position-parallel form under the headline bench
O=gpurun_out/h2ab; mkdir -p $O
for md in bf16x3 f16x2 bf16x3 f16x2; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --matmul-dtype $md > $O/bench_$md.json 2> $O/bench_$md.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$md.json").read().strip().splitlines()[-1])
    r = d["roofline"]; t = r.get("training_step", {})
    print("%-8s value %.4g  ms/update %.2f  bwd %.1f us  fwd %.1f us  gather+fwd+bwd %.1f us  frac %.3f forms %s" % ("$md", d["value"], d["ms_per_step"], r["avg_launch_us"], t.get("forward_kernel_us", 0), t.get("gather_forward_backward_us", 0), r["frac"], d["config"].get("kernel_forms")))
except Exception as e:
    print("$md failed", e); print(open("$O/bench_$md.err").read()[-1500:])
PY
done
