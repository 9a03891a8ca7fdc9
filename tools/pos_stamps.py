"""Phase stamps of cnn_pos_bwd_kernel (workgroup 0, wave 0, fifth super-tile) under the headline launch shape: run with
PQN_T1_STAMPS=1 PQN_BWD_POS=1."""
import ctypes
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys


def main():
    cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", "alg.NUM_ENVS=4096",
                               "alg.TEST_DURING_TRAINING=False"]))
    cfg["MATMUL_DTYPE"] = os.environ.get("MD", "f16x2")
    cfg["TOTAL_TIMESTEPS"] = 8 * 4096 * 32
    train = make_train(dict(cfg), device="cuda:0")
    update, _finish = train.make_batch_runner(seed_keys(0, 16))
    for u in range(3):
        update(u)
    torch.cuda.synchronize()
    lib = _lib.load()
    lib.pqn_debug_pos_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    buf = (ctypes.c_ulonglong * 32)()
    _lib.check(lib.pqn_debug_pos_stamps(buf), "stamps")
    names = ["start", "masks+conv", "conv drain+LN", "dgrad", "drain+LN bwd", "split+dW1", "conv wgrad", "barrier"]
    s = [buf[k] for k in range(len(names))]
    print("pos bwd (one super-tile):", " ".join("%s=%d" % (names[k + 1], s[k + 1] - s[k]) for k in range(len(names) - 1)), "total=%d" % (s[-1] - s[0]))
    f = [buf[16 + k] for k in range(12)]
    fn = ["start", "masks", "conv+LN x4", "split", "fc1", "barrier"]
    print("pos fwd (K step 5):", " ".join("%s=%d" % (fn[k + 1], f[k + 1] - f[k]) for k in range(5)), "step total=%d" % (f[5] - f[0]))
    print("pos fwd tail: head=%d dz planes=%d record=%d" % (f[9] - f[8], f[10] - f[9], f[11] - f[10]))


if __name__ == "__main__":
    main()
