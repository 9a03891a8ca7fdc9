#!/bin/bash
mkdir -p gpurun_out/r2e
timeout 900 python tools/debug_epoch2.py 2048 32 32 2 > gpurun_out/r2e/epoch2_2048.txt 2>&1
grep -n "<<<" gpurun_out/r2e/epoch2_2048.txt | head; sed -n '1,4p;30,40p;60,64p' gpurun_out/r2e/epoch2_2048.txt
