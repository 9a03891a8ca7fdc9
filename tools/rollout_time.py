"""Time of one pqn_cnn_rollout launch (bf16x3 operand mode, T = 32) per MinAtar game at n envs, for the single-tile and
the pair form of the rollout kernel (run under PQN_ROLLOUT_PAIR=0 and =1)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from purejaxql_amd import _lib
from purejaxql_amd.envs import LogWrapper, make
from purejaxql_amd.networks import QNetwork
from purejaxql_amd.profiling import time_launches
from purejaxql_amd.qnet import CnnKernelLayout, cnn_rollout, matmul_mode

GAMES = [("Breakout-MinAtar", 4, 3), ("Asterix-MinAtar", 4, 5), ("Freeway-MinAtar", 7, 3), ("SpaceInvaders-MinAtar", 6, 4)]


def main():
    n, t = int(os.environ.get("N", 65536)), 32
    gpu = torch.device("cuda:0")
    lib = _lib.load()
    for name, c, a in GAMES:
        env, params = make(name, device=gpu)
        env = LogWrapper(env)
        net = QNetwork("cnn", (10, 10, c), a, device=gpu)
        lay = CnnKernelLayout(c, a, matmul_f16=matmul_mode("bf16x3"))
        theta_k = lay.to_kernel(net.init(3))
        (_o, bits0), state = env.reset(11, params, n, want_obs=False, want_bits=True)
        keys = torch.empty(t, dtype=torch.int64, device=gpu)
        _lib.check(lib.pqn_fold_in_range(0x1234567, 7, t, _lib.ptr(keys), _lib.stream_ptr()), "pqn_fold_in_range")
        eps = torch.full((1,), 0.3, dtype=torch.float32, device=gpu)
        words = state.words.clone()
        bits = torch.zeros((t + 1, n, bits0.shape[1]), dtype=bits0.dtype, device=gpu)
        bits[0] = bits0
        ms = time_launches(lambda: cnn_rollout(lay, env._env.env_id, words, bits, theta_k, keys, eps), iters=5)
        print("PQN_ROLLOUT_PAIR=%s %-22s n %d: %.3f ms per rollout" % (os.environ.get("PQN_ROLLOUT_PAIR", "1"), name, n, ms))


if __name__ == "__main__":
    main()
