#!/bin/bash
# round 6 call B: the whole GPU suite (no -x: every failure in one pass)
O=gpurun_out/r6b; mkdir -p $O
timeout 3400 python -m pytest tests/ -q -m gpu > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log; tail -25 $O/pytest_all.log
