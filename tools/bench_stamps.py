"""Phase stamps of the T1 pair kernel under the HEADLINE launch shape (16 seeds x 4096-sample minibatches per launch):
run with PQN_T1_STAMPS=1 PQN_BWD_POS=0 (the position-parallel form that 16-seed launches take by default has its own
stamps: tools/pos_stamps.py).  The single-seed stamps of tools/ablate_train.py come from a half-empty
chip whose 128 workgroups all stream the same seed's planes; these are the phases as the bench sees them."""
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
    cfg["MATMUL_DTYPE"] = "bf16x3"
    cfg["TOTAL_TIMESTEPS"] = 8 * 4096 * 32
    train = make_train(dict(cfg), device="cuda:0")
    update, _finish = train.make_batch_runner(seed_keys(0, 16))
    for u in range(4):
        update(u)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    _lib.check(_lib.load().pqn_debug_t1_stamps(buf), "stamps")
    names = ["start", "inputs", "conv A", "conv B", "fc1 pair + h1T", "act/tgt", "heads A,B", "dgrad A", "P5+P6 A", "dgrad B", "P5+P6 B"]
    for wg in range(4):
        s = [buf[wg * 16 + k] for k in range(len(names))]
        print("WG%d:" % wg, " ".join("%s=%d" % (names[k + 1], s[k + 1] - s[k]) for k in range(len(names) - 1)),
              "total=%d" % (s[-1] - s[0]))


if __name__ == "__main__":
    main()
