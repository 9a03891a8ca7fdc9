#!/bin/bash
# kernel + full-size parity tests, then the headline bench with a kernel trace summary.  Output under gpurun_out/it2/.
mkdir -p gpurun_out/it2
timeout 900 python -m pytest tests/test_qnet_gpu.py tests/test_fullsize_gpu.py tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --steps 10 --warmup 3 --no-extras > gpurun_out/it2/bench.json 2> gpurun_out/it2/bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/it2/bench.json").read().strip().splitlines()[-1])
print("value %.4g  ms/step %.2f  T1 us %.1f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pi
rocprofv3 --kernel-trace -d /tmp/pi -o x -- python $R/bench.py --steps 6 --warmup 2 --no-extras > /dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/pi/x_results.db 8 | cut -c1-150 | tee $R/gpurun_out/it2/kernel_stats.txt
