import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pqn_oracle as O
from purejaxql_amd.envs import FlattenObservationWrapper, LogWrapper, make
f = np.float32
def sincos(x):
    k = np.rint(x * f(0.636619772367581343))
    r = x - k * f(1.5703125); r = r - k * f(4.837512969970703125e-4); r = r - k * f(7.54978995489188216e-8)
    z = r * r
    sp = ((f(-1.9515295891e-4) * z + f(8.3321608736e-3)) * z - f(1.6666654611e-1)) * z * r + r
    cp = ((f(2.443315711809948e-5) * z - f(1.388731625493765e-3)) * z + f(4.166664568298827e-2)) * z * z - f(0.5) * z + f(1.0)
    return sp, cp   # |x| < pi/4 here
def step_np(s, a):
    x, xd, th, thd = (s[:, i].copy() for i in range(4))
    force = f(10.0) * a.astype(f) - f(10.0) * (1 - a).astype(f)
    sn, cs = sincos(th)
    temp = (force + f(0.05) * (thd * thd) * sn) / f(1.1)
    thacc = (f(9.8) * sn - cs * temp) / (f(0.5) * (f(4.0) / f(3.0) - f(0.1) * (cs * cs) / f(1.1)))
    xacc = temp - f(0.05) * thacc * cs / f(1.1)
    return np.stack([x + f(0.02) * xd, xd + f(0.02) * xacc, th + f(0.02) * thd, thd + f(0.02) * thacc], 1).astype(f)
dev = torch.device("cuda:0")
env, params = make("CartPole-v1", device=dev)
env = LogWrapper(FlattenObservationWrapper(env))
oenv = O.OracleEnv("CartPole-v1")
n = 512
obs, state = env.reset(3, params, n)
oobs, ost = oenv.reset(3, n)
rng = np.random.default_rng(0)
for t in range(60):
    a = rng.integers(0, 2, n).astype(np.int32)
    prev = obs.cpu().numpy()
    ref = step_np(prev, a)
    obs, state, r, d, info = env.step(900 + t, state, torch.from_numpy(a).to(dev), params)
    pso = oobs.copy()
    oobs, ost, orr, od, oinfo = oenv.step(900 + t, ost, a)
    g = obs.cpu().numpy()
    nd = ~(d.cpu().numpy().astype(bool))
    dg = (g[nd].view(np.int32) - ref[nd].view(np.int32))
    refo = step_np(pso, a)
    ndo = ~od
    do = (oobs[ndo].view(np.int32) - refo[ndo].view(np.int32))
    print(t, "gpu vs numpy-f32 ulp diff per column max", np.abs(dg).max(0), "| oracle vs numpy-f32", np.abs(do).max(0), "| gpu==oracle", np.array_equal(g, oobs))
    if not np.array_equal(g, oobs):
        break
