"""per-tensor error of the wide-MLP gradient vs the oracle (C5 shape)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import pqn_oracle as O
from purejaxql_amd.networks import QNetwork
from purejaxql_amd.qnet import BigMlpKernelLayout, BigMlpTrainer
gpu = torch.device("cuda:0")
d, h, layers, a, nb = (1345, 1024, 4, 17, 1024) if len(sys.argv) < 2 else tuple(int(x) for x in sys.argv[1:6])
norm_input = renorm = True
torch.manual_seed(11)
net = QNetwork("mlp", (d,), a, norm_type="layer_norm", norm_input=norm_input, hidden_size=h, num_layers=layers, device=gpu, renorm=renorm)
lay = BigMlpKernelLayout(d, h, layers, a, 2)
theta = net.init(11) + 0.03 * torch.randn(net.num_params, device=gpu)
tr = BigMlpTrainer(lay, theta, 1e-4, 1.0, lr_decay_steps=500.0)
shapes = O.mlp_shapes(d, a, h, layers, "layer_norm", renorm)
p = O.unflatten(theta.cpu().numpy(), shapes)
rng = np.random.default_rng(nb + d)
n_env = nb // 2
rows = 3 * n_env
obs_all = (rng.standard_normal((rows, d)) * (rng.random(d) * 1.5) + 0.3 * rng.standard_normal(d)).astype(np.float32)
action = rng.integers(0, a, nb).astype(np.int32); reward = rng.standard_normal(nb).astype(np.float32); done = rng.random(nb) < 0.2
idx = rng.permutation(nb).astype(np.int64)
stats = {"BatchRenorm_0/mean": np.zeros(d, np.float32), "BatchRenorm_0/var": np.ones(d, np.float32), "BatchRenorm_0/steps": 3}
tr.in_steps[0] = 3
g = tr.compute_grad(torch.from_numpy(idx).to(gpu), torch.from_numpy(obs_all).to(gpu), torch.from_numpy(action).to(gpu),
                    reward=torch.from_numpy(reward).to(gpu), done=torch.from_numpy(done).to(gpu), gamma=0.99, next_offset=n_env)
lo, chosen, g_ref = O.net_loss_grad_1step("mlp", p, shapes, obs_all[idx], obs_all[idx + n_env], action[idx], reward[idx], done[idx], 0.99,
                                          layers=layers, norm_type="layer_norm", norm_input=True, stats=stats, new_stats={}, renorm=True)
gf = lay.to_flax(g).cpu().numpy()
off = 0
for k, s in shapes.items():
    n = int(np.prod(s))
    r, x = g_ref[off:off + n], gf[off:off + n]
    e = np.abs(x - r)
    bad = e > 2e-3 * np.abs(r) + 1e-5 * np.abs(g_ref).max()
    msg = ""
    if bad.any() and len(s) == 2:
        bb = bad.reshape(s)
        msg = f" bad rows {int(bb.any(1).sum())}/{s[0]} cols {int(bb.any(0).sum())}/{s[1]} first bad {np.argwhere(bb)[:3].tolist()}"
    print(f"{k:24s} max|ref| {np.abs(r).max():.3e} max err {e.max():.3e} rel-L2 {np.linalg.norm(x - r) / max(np.linalg.norm(r), 1e-30):.2e} bad {int(bad.sum())}{msg}")
    off += n

# relu-mask agreement between the kernel's forward (activations in the workspace) and the oracle's forward
xx = np.concatenate((obs_all[idx], obs_all[idx + n_env]))
q_all, cache = O.net_forward("mlp", p, xx, layers=layers, want_cache=True, norm_type="layer_norm", norm_input=True, train=True,
                             stats=stats, new_stats={}, renorm=True)
print("xn max|diff|", np.abs(tr.intermediate(2 * nb, nb, "xn").cpu().numpy()[:, :d] - cache["hs"][0]).max())
for l in range(layers):
    hk = tr.intermediate(2 * nb, nb, "h", l).cpu().numpy()
    ho = cache["hs"][l + 1]
    mism = (hk > 0) != (ho > 0)
    print(f"layer {l}: max|h diff| {np.abs(hk - ho).max():.3e}  relu-mask mismatches {int(mism.sum())} (gradient rows: {int(mism[:nb].sum())}) "
          f"max |h| at a mismatch {float(np.maximum(np.abs(hk), np.abs(ho))[mism].max()) if mism.any() else 0:.3e}")
