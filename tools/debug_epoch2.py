"""Side-by-side SGD trajectory: the oracle's optimizer loop and the fused HIP kernels on the SAME rollout data (taken from
the oracle) and the SAME permutations, compared after every optimizer step: where do the two part, and how (one hidden
unit at a time = a discrete ReLU-gate event, or a smooth drift)?  usage: debug_epoch2.py N T MB EP"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pqn_oracle as O
from purejaxql_amd.networks import QNetwork
from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer

N, T, MB, EP = [int(x) for x in sys.argv[1:5]]
dev = torch.device("cuda:0")
env = O.OracleEnv("Breakout-MinAtar")
net = QNetwork("cnn", (10, 10, 4), 3, device=dev)
theta0 = net.init(123)
shapes = O.cnn_shapes((10, 10, 4), 3)
th = theta0.cpu().numpy().copy()
p = O.unflatten(th, shapes)
# rollout with uniformly random actions (eps = 1 at update 0) through the oracle
obs, st = env.reset(7, N)
Obs = np.zeros((T + 1, N, 10, 10, 4), np.float32); Obs[0] = obs
A = np.zeros((T, N), np.int32); R = np.zeros((T, N), np.float32); D = np.zeros((T, N), bool); QM = np.zeros((T, N), np.float32)
for t in range(T):
    q = O.net_forward("cnn", p, Obs[t])
    A[t], QM[t] = O.eps_greedy(q, np.float32(1.0), 100 + t)
    Obs[t + 1], st, R[t], D[t], _ = env.step(100 + t, st, A[t])
last_q = O.net_forward("cnn", p, Obs[T]).max(-1)
tgt = O.q_lambda(R, D, QM, last_q, 0.99, 0.65)
of, af, tf = Obs[:T].reshape(T * N, 10, 10, 4), A.reshape(-1), tgt.reshape(-1)
flat = of.reshape(T * N, -1).astype(np.uint64)
padded = np.zeros((T * N, 512), np.uint64); padded[:, :400] = flat
words = (padded.reshape(-1, 16, 32) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
bits = torch.from_numpy(words.view(np.int32)).to(dev)
act_t, tgt_t = torch.from_numpy(af).to(dev), torch.from_numpy(tf).to(dev)
lay = CnnKernelLayout(4, 3)
B = T * N // MB
lr_steps = 30 * MB * EP
tr = CnnTrainer(lay, theta0, 5e-4, 10.0, lr_decay_steps=float(lr_steps), max_minibatch=B)       # free-running
ts = CnnTrainer(lay, theta0, 5e-4, 10.0, lr_decay_steps=float(lr_steps), max_minibatch=B)       # re-synchronised to the oracle's theta
m, v = np.zeros_like(th), np.zeros_like(th)
off, n = net.offsets["CNN_0/Dense_0/kernel"]
step = 0
prev = 0.0
for ep in range(EP):
    perm = O.permutation(O.fold_in(99, ep), T * N)
    for mb in range(MB):
        idx = perm[mb * B:(mb + 1) * B]
        loss, chosen, g = O.net_loss_grad("cnn", p, shapes, of[idx], af[idx], tf[idx])
        idx_t = torch.from_numpy(idx.astype(np.int64)).to(dev)
        ts.theta.copy_(lay.to_kernel(torch.from_numpy(th).to(dev)))
        lay.refresh_copies(ts.theta, ts.w1b)
        g_gpu = lay.to_flax(ts.compute_grad(idx_t, bits, act_t, tgt_t)).cpu().numpy()     # same theta as the oracle
        tr.compute_grad(idx_t, bits, act_t, tgt_t)
        gd = np.abs(g_gpu - g)
        gk = gd[off:off + n].reshape(1024, 128)
        col_err = gk.max(0) / (np.abs(g[off:off + n]).reshape(1024, 128).max(0) + 1e-30)
        worst = int(col_err.argmax())
        tr.apply()
        lr = O.linear_schedule(5e-4, 1e-20, lr_steps, step)
        O.radam_clip_step(th, g, m, v, step, np.float32(lr), 10.0)
        thg = tr.theta_flax().cpu().numpy()
        rel = np.linalg.norm(thg - th) / np.linalg.norm(th - theta0.cpu().numpy())
        flag = " <<<" if rel > 3 * max(prev, 1e-6) else ""
        print(f"ep {ep} mb {mb:3d} step {step:3d}: grad max rel err {gd.max() / np.abs(g).max():.2e}, worst fc1 column {worst} "
              f"(col rel err {col_err[worst]:.2e}); theta rel diff of update {rel:.3e}{flag}", flush=True)
        prev = rel
        step += 1
