#!/bin/bash
# full GPU check: the whole -m gpu suite, smoke(), then the default bench line
mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/full/bench.json").read().strip().splitlines()[-1])
print("value %.4g  ms/step %.2f dtype %s" % (d["value"], d["ms_per_step"], d["dtype"]))
print("roofline", json.dumps(d["roofline"])[:400])
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
for k in ("single_seed", "matmul_modes"):
    if k in d: print(k, json.dumps(d[k])[:500])
PY
