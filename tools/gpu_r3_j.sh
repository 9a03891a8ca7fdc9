#!/bin/bash
# Round 3, call j: the whole -m gpu suite, smoke(), and the default bench line.
mkdir -p gpurun_out/r3j
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r3j/pytest.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r3j/pytest.txt
tail -30 gpurun_out/r3j/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3j/smoke.txt 2>&1
tail -3 gpurun_out/r3j/smoke.txt
timeout 600 python bench.py > gpurun_out/r3j/bench.json 2> gpurun_out/r3j/bench.err
tail -c 3000 gpurun_out/r3j/bench.json
