#!/bin/bash
# round 3, job A: the new headline-configuration tests, the in-process option tests, C5-shape tests, bench self-launch
mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
timeout 900 python -m pytest tests/test_headline_gpu.py tests/test_qnet_gpu.py -x -q -m gpu -k "headline or forced or 8192 or opt_in" > $O/pytest_headline.log 2>&1
tail -15 $O/pytest_headline.log
timeout 900 python -m pytest tests/test_craftax_gpu.py tests/test_craftax_env_gpu.py -q -m gpu > $O/pytest_craftax.log 2>&1
tail -15 $O/pytest_craftax.log
PQN_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-extras > $O/bench_gpus2.json 2> $O/bench_gpus2.err
tail -c 600 $O/bench_gpus2.json; tail -5 $O/bench_gpus2.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3a/bench.json").read().strip().splitlines()[-1])
print("value %.4g ms/step %.2f T1 us %.1f frac %.3f forms %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["config"].get("kernel_forms")))
print("sustained", d.get("sustained")); print("cpu", d.get("cpu_baseline")); print("single", d.get("single_seed", {}).get("value"))
print("env_step", [(r.get("num_envs"), round(r.get("achieved", 0)), r.get("level", "")[:20]) for r in d.get("roofline_env_step", []) if isinstance(r, dict)])
PY
tail -3 $O/bench.err
