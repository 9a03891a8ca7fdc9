import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from purejaxql_amd import _lib
from purejaxql_amd.networks import QNetwork
from purejaxql_amd.qnet import CnnKernelLayout, cnn_grad_seeds, matmul_mode
import importlib
oracle = importlib.import_module("pqn_oracle")
gpu = torch.device("cuda:0")
S = 16
def pack(obs):
    n, c = obs.shape[0], obs.shape[-1]
    ow = (((100 * c + 31) // 32) + 3) // 4 * 4
    padded = np.zeros((n, ow * 32), np.uint64)
    padded[:, :100 * c] = obs.reshape(n, -1)
    return (padded.reshape(n, ow, 32) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
c, a, stacked = 6, 4, False
rng = np.random.default_rng(2026 + stacked + 10 * c)
torch.manual_seed(7)
nb = 4096
n_env, t_len = 20000, 1
rows = n_env
obs = (rng.random((rows, 10, 10, c)) < 0.12).astype(np.float32)
bits = torch.from_numpy(pack(obs).view(np.int32)).to(gpu)
action = rng.integers(0, a, rows).astype(np.int32)
target = rng.standard_normal(rows).astype(np.float32)
net = QNetwork("cnn", (10, 10, c), a, device=gpu)
lay = CnnKernelLayout(c, a, matmul_f16=matmul_mode("bf16x3"))
stride = (lay.alloc + 3) // 4 * 4
thetas = [net.init(100 + s) + 0.05 * torch.randn(net.num_params, device=gpu) for s in range(S)]
theta_k = torch.zeros((S, stride), dtype=torch.float32, device=gpu)
for s in range(S):
    theta_k[s, :lay.alloc] = lay.to_kernel(thetas[s])
idx = np.stack([rng.permutation(n_env * t_len)[:nb] for _ in range(S)]).astype(np.int64)
res = {}
for form, opt in (("pos", 1), ("pair", 0)):
    with _lib.options(bwd_pos=opt):
        grad, loss, qv = cnn_grad_seeds(lay, theta_k, torch.from_numpy(idx).to(gpu), bits, torch.from_numpy(action).to(gpu),
                                        torch.from_numpy(target).to(gpu), n_env, n_env)
        print(form, _lib.last_kernel_form())
    res[form] = grad.clone()
shapes = oracle.cnn_shapes((10, 10, c), a)
names = list(shapes.keys()); sizes = [int(np.prod(shapes[k])) for k in names]
offs = np.cumsum([0] + sizes)
for s in (8, 9, 10):
    j = idx[s]
    p = oracle.unflatten(thetas[s].cpu().numpy(), shapes)
    lo, chosen, g_ref = oracle.net_loss_grad("cnn", p, shapes, obs[j], action[j], target[j])
    for form in ("pos", "pair"):
        g = lay.to_flax(res[form][s]).cpu().numpy()
        d = np.abs(g - g_ref)
        bad = d > (2e-3 * np.abs(g_ref) + 3e-6 * np.abs(g_ref).max() + 1e-9)
        print("seed", s, form, "bad", int(bad.sum()), "max abs", float(d.max()))
        for k, (o0, o1) in zip(names, zip(offs[:-1], offs[1:])):
            nbad = int(bad[o0:o1].sum())
            if nbad:
                ii = np.nonzero(bad[o0:o1])[0]
                print("   ", k, shapes[k], "bad", nbad, "max abs", float(d[o0:o1].max()), "first idx", ii[:12], "ref", g_ref[o0:o1][ii[:4]], "got", g[o0:o1][ii[:4]])
