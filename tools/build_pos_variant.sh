#!/bin/bash
# build purejaxql_amd/csrc/variants/libpqn_hip_<name>.so = the default library with pqn_qnet_pos.hip recompiled with extra -D flags
# usage: tools/build_pos_variant.sh <name> [-DFLAG ...]
set -e
cd "$(dirname "$0")/../purejaxql_amd/csrc"
name=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -fno-slp-vectorize "$@" -c pqn_qnet_pos.hip -o /tmp/pqn_qnet_pos_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libpqn_hip_$name.so pqn_env.o pqn_algo.o pqn_qnet.o /tmp/pqn_qnet_pos_$name.o pqn_update.o pqn_mlp.o pqn_craftax.o pqn_bigmlp.o pqn_peer.o
echo built variants/libpqn_hip_$name.so
