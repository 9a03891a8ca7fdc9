#!/bin/bash
# round 6: existing kernel-selection options forced at the launch shapes below the position-parallel form's threshold (is any default rule leaving time on the table?)
run() { echo -n "$1 | "; shift; env "$@" timeout 200 python tools/shape_run.py $SHAPE 2>/dev/null | tail -1; }
SHAPE="4096 1 30"
run "default" X=1
run "t1_pair=2" PQN_T1_PAIR=2
run "t2_acc=2" PQN_T2_ACC=2
run "t1_pair=2 t2_acc=2" PQN_T1_PAIR=2 PQN_T2_ACC=2
SHAPE="4096 1 30 f16x2"
run "f16x2 pos forced, 2 waves / 8 chunks" PQN_BWD_POS=2 PQN_ROLLOUT_POS=2 PQN_POS_WAVES=2 PQN_POS_CHUNKS=8
run "f16x2 pos train only, 2 waves / 8 chunks" PQN_BWD_POS=2 PQN_POS_WAVES=2 PQN_POS_CHUNKS=8
run "f16x2 pos rollout only, 2 waves" PQN_ROLLOUT_POS=2 PQN_POS_WAVES=2
SHAPE="1024 1 60"
run "default" X=1
run "t1_pair=2" PQN_T1_PAIR=2
run "t2_acc=2" PQN_T2_ACC=2
run "rollout_pair=2" PQN_ROLLOUT_PAIR=2
SHAPE="1024 1 60 f32"
run "f32" X=1
SHAPE="128 1 100"
run "default" X=1
run "t1_ksplit=2" PQN_T1_KSPLIT=2
run "t1_ksplit=3" PQN_T1_KSPLIT=3
run "t1_ksplit=4" PQN_T1_KSPLIT=4
SHAPE="128 1 100 bf16x3"
run "bf16x3" X=1
run "bf16x3 rollout_pair=2" PQN_ROLLOUT_PAIR=2
SHAPE="128 1 100 f16"
run "f16" X=1
