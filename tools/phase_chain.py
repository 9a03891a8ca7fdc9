"""per-kernel cost of the yaml-default training step's kernels inside a torch-captured graph, by phase: N x GRAD (ks_fwd, ks_hb, reduce),
N x APPLY (radam_norm, radam_apply), N x (GRAD, APPLY)"""
import sys, os, time, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys
n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lib = _lib.load()
cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", f"alg.NUM_ENVS={n_envs}", "alg.TEST_DURING_TRAINING=False"]))
cfg["TOTAL_TIMESTEPS"] = 20 * n_envs * 32
tr = make_train(cfg, device="cuda:0")
upd, _ = tr.make_runner(seed_keys(0, 1)[0])
for u in range(3):
    upd(u)
torch.cuda.synchronize()
drv = upd.driver
a = drv.args
st = torch.cuda.Stream()
N = 512
def chain(name, phases, kernels_per_iter):
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for i in range(N):
                for ph in phases:
                    _lib.check(lib.pqn_cnn_update_phase(C.byref(a), ph, i % 64, _lib.stream_ptr()), "phase")
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (4 * N) * 1e6
    print("%-28s %.2f us per iteration (%d kernels: %.2f us each)  forms %s" % (name, dt, kernels_per_iter, dt / kernels_per_iter, _lib.last_kernel_form()), flush=True)
ks = n_envs <= 256
chain("GRAD", [2], 3)
chain("APPLY (norm + apply)", [3], 2)
chain("GRAD + APPLY", [2, 3], 5)
