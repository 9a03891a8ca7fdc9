set -x
timeout 300 python -m purejaxql_amd.pqn_minatar alg.ENV_NAME=Breakout-MinAtar alg.NUM_ENVS=4096 NUM_SEEDS=16 alg.MATMUL_DTYPE=bf16x3 alg.TOTAL_TIMESTEPS=2e7 alg.TOTAL_TIMESTEPS_DECAY=2e7 SAVE_PATH=/tmp/ckpt 2>&1 | grep -v amdgpu.ids | tail -3; ls /tmp/ckpt/Breakout-MinAtar | wc -l
timeout 300 python tools/learn_headline.py Freeway-MinAtar 1e7 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python tools/learn_headline.py SpaceInvaders-MinAtar 1e7 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python -m purejaxql_amd.pqn_craftax alg.ENV_NAME=Craftax-Classic-Symbolic-v1 alg.TOTAL_TIMESTEPS=3e6 alg.TOTAL_TIMESTEPS_DECAY=3e6 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python -m purejaxql_amd.pqn_minatar alg.ENV_NAME=Asterix-MinAtar NUM_SEEDS=3 alg.TOTAL_TIMESTEPS=3e6 alg.TOTAL_TIMESTEPS_DECAY=3e6 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python -m purejaxql_amd.pqn_gymnax +alg=pqn_cartpole NUM_SEEDS=4 2>&1 | grep -v amdgpu.ids | tail -2
