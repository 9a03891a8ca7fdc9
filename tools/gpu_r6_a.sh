#!/bin/bash
# round 6: position-form parity tests, then the headline A/B of the current library against prebuilt variants ($@)
O=gpurun_out/r6a; mkdir -p $O
timeout 1200 python -m pytest tests/test_qnet_gpu.py -x -q -m gpu -k "position" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
bash tools/gpu_pos_ab.sh "$@" 2>&1 | tee $O/ab.txt
