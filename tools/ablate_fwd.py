import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from purejaxql_amd.networks import QNetwork
from purejaxql_amd.qnet import CnnKernelLayout, cnn_forward
from purejaxql_amd.profiling import time_launches
from purejaxql_amd.envs import make
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env, params = make("Breakout-MinAtar", device=dev)
(obs, bits), state = env.reset(3, params, n, want_bits=True)
for t in range(30):
    (obs, bits), state, *_ = env.step(t, state, torch.randint(0, 3, (n,), dtype=torch.int32, device=dev), params, want_bits=True)
net = QNetwork("cnn", (10, 10, 4), 3, device=dev); lay = CnnKernelLayout(4, 3)
th = lay.to_kernel(net.init(0))
q = torch.empty((n, 3), device=dev); a = torch.empty(n, dtype=torch.int32, device=dev); qm = torch.empty(n, device=dev)
ms = time_launches(lambda: cnn_forward(lay, bits, th, eps=0.1, key=5, q=q, action=a, qmax=qm), iters=300)
print("PQN_ABLATE=%s n=%d fwd kernel %.2f us" % (os.environ.get("PQN_ABLATE", "0"), n, ms * 1e3))
