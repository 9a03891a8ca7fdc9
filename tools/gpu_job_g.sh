#!/bin/bash
mkdir -p gpurun_out/r2g
timeout 300 python tools/debug_cartpole.py > gpurun_out/r2g/cartpole.txt 2>&1
tail -8 gpurun_out/r2g/cartpole.txt
