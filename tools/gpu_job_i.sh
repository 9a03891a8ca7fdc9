#!/bin/bash
mkdir -p gpurun_out/r2i
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r2i/pytest.txt 2>&1
echo "rc=$?" >> gpurun_out/r2i/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2i/bench.json 2> gpurun_out/r2i/bench.err
tail -12 gpurun_out/r2i/pytest.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2i/bench.json').read())
print("value", d["value"], "T1 us", d["roofline"]["avg_launch_us"], "modes", json.dumps(d.get("matmul_modes")), "single", d.get("single_seed",{}).get("value"), d.get("single_seed",{}).get("roofline",{}).get("avg_launch_us"))
PY
