"""Digest of pqn_cnn_rollout's outputs (bf16x3 operand mode) on fixed inputs, one line per case.  The rollout kernel has a
single-tile and a pair form (PQN_ROLLOUT_PAIR=0 / 2, read when the library loads); tests/test_qnet_gpu.py runs this
script under both and requires identical output."""
import hashlib
import torch
from purejaxql_amd import _lib
from purejaxql_amd.envs import LogWrapper, make
from purejaxql_amd.networks import QNetwork
from purejaxql_amd.qnet import CnnKernelLayout, cnn_rollout, matmul_mode

CASES = [("Breakout-MinAtar", 4, 3, 64, 24), ("Breakout-MinAtar", 4, 3, 1024, 12), ("Asterix-MinAtar", 4, 5, 96, 16),
         ("Freeway-MinAtar", 7, 3, 64, 12), ("SpaceInvaders-MinAtar", 6, 4, 32, 20)]


def main():
    gpu = torch.device("cuda:0")
    lib = _lib.load()
    for name, c, a, n, t in CASES:
        env, params = make(name, device=gpu)
        env = LogWrapper(env)
        net = QNetwork("cnn", (10, 10, c), a, device=gpu)
        lay = CnnKernelLayout(c, a, matmul_f16=matmul_mode("bf16x3"))
        torch.manual_seed(0)
        theta_k = lay.to_kernel(net.init(3) + 0.05 * torch.randn(net.num_params, device=gpu))
        (_o, bits0), state = env.reset(11, params, n, want_obs=False, want_bits=True)
        for i in range(30):
            (_o, bits0), state, *_ = env.step(500 + i, state, torch.randint(0, a, (n,), dtype=torch.int32, device=gpu),
                                              params, want_obs=False, want_bits=True)
        keys = torch.empty(t, dtype=torch.int64, device=gpu)
        _lib.check(lib.pqn_fold_in_range(0x1234567, 7, t, _lib.ptr(keys), _lib.stream_ptr()), "pqn_fold_in_range")
        eps = torch.full((1,), 0.3, dtype=torch.float32, device=gpu)
        words = state.words.clone()
        bits = torch.zeros((t + 1, n, bits0.shape[1]), dtype=bits0.dtype, device=gpu)
        bits[0] = bits0
        rec = cnn_rollout(lay, env._env.env_id, words, bits, theta_k, keys, eps)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        parts = []
        for k, v in sorted(rec.items()) + [("words", words), ("bits", bits)]:
            b = v.cpu().numpy().tobytes()
            h.update(b)
            parts.append("%s:%s" % (k, hashlib.sha256(b).hexdigest()[:8]))
        print(name, n, t, h.hexdigest(), " ".join(parts), "dones", int(rec["done"].sum()))


if __name__ == "__main__":
    main()
