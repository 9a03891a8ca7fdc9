import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from purejaxql_amd.networks import QNetwork
from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer
from purejaxql_amd.profiling import time_launches
from purejaxql_amd.envs import make
dev = torch.device("cuda:0")
n, T = 4096, 8
env, params = make("Breakout-MinAtar", device=dev)
allbits = []
(obs, bits), state = env.reset(3, params, n, want_bits=True, want_obs=False)
for t in range(T + 30):
    (obs, bits), state, *_ = env.step(t, state, torch.randint(0, 3, (n,), dtype=torch.int32, device=dev), params, want_bits=True, want_obs=False)
    if t >= 30: allbits.append(bits)
bits = torch.cat(allbits)
net = QNetwork("cnn", (10, 10, 4), 3, device=dev); lay = CnnKernelLayout(4, 3, matmul_f16=int(os.environ.get("PQN_MODE", os.environ.get("PQN_F16", "0"))))   # 0 f32, 1 f16, 2 bf16x3
tr = CnnTrainer(lay, net.init(0), 5e-4, 10.0)
idx = torch.randperm(n * T, device=dev)[:4096].contiguous()
act = torch.randint(0, 3, (n * T,), dtype=torch.int32, device=dev); tgt = torch.randn(n * T, device=dev)
ms = time_launches(lambda: tr.compute_grad(idx, bits, act, tgt), iters=200)
ms2 = time_launches(lambda: tr.apply(), iters=200)
print("PQN_ABLATE_TRAIN=%s grad(T1+T2+T3a) %.2f us  apply %.2f us" % (os.environ.get("PQN_ABLATE_TRAIN", "0"), ms * 1e3, ms2 * 1e3))

if os.environ.get("PQN_T1_STAMPS"):
    import ctypes
    from purejaxql_amd import _lib
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    _lib.check(_lib.load().pqn_debug_t1_stamps(buf), "stamps")
    pair = buf[10] != 0 and os.environ.get("PQN_T1_PAIR", "1") != "0" and int(os.environ.get("PQN_MODE", "0")) == 2
    if pair:   # qnet_cnn_train_pair_kernel: two tiles per workgroup
        names = ["start", "inputs", "conv A", "conv B", "fc1 pair", "h1T x2 + mask", "heads A,B", "dgrad A", "P5+P6 A", "dgrad B", "P5+P6 B"]
    else:
        names = ["start", "inputs", "conv(phase1)", "h1T+fc1", "head+bwd+dzT", "dgrad", "P5 ln0 bwd", "P6 conv wgrad"]
    n = len(names)
    for wg in range(4):
        s = [buf[wg * 16 + k] for k in range(n)]
        print("WG%d cycles:" % wg, " ".join("%s=%d" % (names[k + 1], s[k + 1] - s[k]) for k in range(n - 1)), "total=%d" % (s[n - 1] - s[0]))
