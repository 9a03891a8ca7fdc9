#!/bin/bash
# phase stamps of the position-parallel kernels (variant library built with -DPOS_STAMPS) under the headline launch shape
O=gpurun_out/r5s; mkdir -p $O
L=purejaxql_amd/csrc/libpqn_hip.so
cp $L /tmp/libpqn_default.so
cp purejaxql_amd/csrc/variants/libpqn_hip_stamps.so $L
PQN_BWD_POS=1 PQN_T1_STAMPS=1 timeout 300 python tools/pos_stamps.py > $O/stamps.txt 2>&1; tail -4 $O/stamps.txt
cp /tmp/libpqn_default.so $L
