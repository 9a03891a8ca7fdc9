#!/bin/bash
mkdir -p gpurun_out/r2d
timeout 900 python tools/debug_e2e_matrix.py 4096,32,32,2 4096,32,32,1 4096,32,64,2 2048,32,32,2 4096,16,32,2 4096,32,8,1 > gpurun_out/r2d/matrix.txt 2>&1
timeout 1500 python -m pytest tests/test_qnet_gpu.py tests/test_craftax_gpu.py -q -x > gpurun_out/r2d/pytest_qnet.txt 2>&1
echo "rc=$?" >> gpurun_out/r2d/pytest_qnet.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -q -k "end_to_end and not 4096" > gpurun_out/r2d/pytest_e2e.txt 2>&1
echo "rc=$?" >> gpurun_out/r2d/pytest_e2e.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err
cat gpurun_out/r2d/matrix.txt; tail -15 gpurun_out/r2d/pytest_qnet.txt; tail -8 gpurun_out/r2d/pytest_e2e.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2d/bench.json').read())
print("value", d["value"], "T1 us", d["roofline"]["avg_launch_us"], "modes", json.dumps(d.get("matmul_modes")), "single", d.get("single_seed",{}).get("value"))
PY
