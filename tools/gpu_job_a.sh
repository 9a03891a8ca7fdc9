#!/bin/bash
# round-2 GPU job A: micro-benchmarks, the whole -m gpu suite, a short bench line
mkdir -p gpurun_out/r2a
(cd tools/ubench && ./bf16x3 > ../../gpurun_out/r2a/ubench.txt 2>&1)
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
echo "bench rc=$?" >> gpurun_out/r2a/bench.err
tail -5 gpurun_out/r2a/pytest.txt; cat gpurun_out/r2a/ubench.txt; head -c 1500 gpurun_out/r2a/bench.json
