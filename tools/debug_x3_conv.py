"""T1 in bf16x3 mode vs f32 mode on identical inputs, repeated: run-to-run determinism and per-block differences."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_qnet_gpu import _random_bits
from purejaxql_amd.networks import QNetwork
from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer
dev = torch.device("cuda:0")
for (c, a, nb, pool) in [(4, 3, 16, 64), (4, 3, 128, 1000), (4, 3, 4096, 20000), (6, 4, 1024, 2000), (7, 3, 1024, 1024), (10, 6, 1024, 2000)]:
    rng = np.random.default_rng(nb + c); torch.manual_seed(1234)
    net = QNetwork("cnn", (10, 10, c), a, device=dev)
    theta = net.init(11) + 0.05 * torch.randn(net.num_params, device=dev)
    obs, words = _random_bits(rng, pool, c, density=0.12)
    bits = torch.from_numpy(words.view(np.int32)).to(dev)
    action = torch.from_numpy(rng.integers(0, a, pool).astype(np.int32)).to(dev)
    target = torch.from_numpy(rng.standard_normal(pool).astype(np.float32)).to(dev)
    idx = torch.from_numpy((rng.permutation(pool)[:nb]).astype(np.int64)).to(dev)
    if os.environ.get("BRIEF"): print("C", c, "nb", nb, end=": ")
    res = {}
    for mode in (0, 2):
        lay = CnnKernelLayout(c, a, matmul_f16=mode)
        tr = CnnTrainer(lay, theta, 5e-4, 10.0, lr_decay_steps=1000.0)
        outs = []
        for rep in range(3):
            lo = torch.zeros(1, device=dev); qv = torch.zeros(1, device=dev)
            g = tr.compute_grad(idx, bits, action, target, lo, qv)[:lay.total].clone()
            outs.append((float(lo), float(qv), g))
        res[mode] = (lay, outs)
        if os.environ.get("BRIEF"):
            if mode == 2:
                d02 = float((res[0][1][0][2] - outs[0][2]).abs().max())
                print("qv f32 %.7f x3 %s rep-to-rep max|dg| %.2e  max|g_x3 - g_f32| %.2e" % (res[0][1][0][1], [round(o[1], 7) for o in outs], max(float((outs[0][2] - o[2]).abs().max()) for o in outs[1:]), d02))
            continue
        print("nb", nb, "mode", mode, "loss/qv per rep:", [(round(o[0], 7), round(o[1], 7)) for o in outs],
              "rep-to-rep max|dg|", max(float((outs[0][2] - o[2]).abs().max()) for o in outs[1:]))
    if os.environ.get("BRIEF"): continue
    lay, o2 = res[2]; _, o0 = res[0]
    d = (o2[0][2] - o0[0][2]).abs()
    names = [n for n in dir(lay.struct) if n.startswith("off_")]
    offs = sorted((getattr(lay.struct, n), n) for n in names if getattr(lay.struct, n) < lay.total)
    for k, (o, n) in enumerate(offs):
        e = offs[k + 1][0] if k + 1 < len(offs) else lay.total
        print("   %-10s max|d| %.3e  max|g| %.3e" % (n, float(d[o:e].max()), float(o0[0][2][o:e].abs().max())))
