#!/bin/bash
# Round 3, call o: env-sharded mode with two ranks on the one GPU (gloo process group): peer all-reduce inside one graph
# vs the host-issued collective between 65 graph segments.  Absolute rates mean nothing (both ranks share the GPU).
mkdir -p gpurun_out/r3o
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PQN_BENCH_ONE_GPU=1 PQN_DIST_BACKEND=gloo
for peer in 1 0; do
  PQN_PEER_ALLREDUCE=$peer timeout 600 python bench.py --gpus 2 --mode envs --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r3o/envs_peer$peer.json 2> gpurun_out/r3o/envs_peer$peer.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3o/envs_peer$peer.json"))
    print("peer=$peer", d["value"], d["ms_per_step"], d["config"].get("grad_allreduce"), d["config"].get("driver"), d["n_gpus"])
except Exception as e:
    print("peer=$peer failed", repr(e)); print(open("gpurun_out/r3o/envs_peer$peer.err").read()[-1500:])
PY
done
