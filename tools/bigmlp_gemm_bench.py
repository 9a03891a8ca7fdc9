"""times the wide-MLP GEMM kernel at one shape: python tools/bigmlp_gemm_bench.py M N K tile nsplit [reps]
(run under rocprofv3 --kernel-trace --stats: the bm_gemm_kernel row is the number; split / transpose are the operand prep)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from purejaxql_amd import _lib
lib = _lib.load()
m, n, k, tile, nsplit = (int(x) for x in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 50
gpu = torch.device("cuda:0")
a = torch.randn(m, k, device=gpu); b = torch.randn(n, k, device=gpu)
c = torch.empty(nsplit, m, n, device=gpu)
scratch = torch.empty(int(lib.pqn_bigmlp_gemm_scratch_floats(m, n, k)), device=gpu)
for _ in range(reps):
    _lib.check(lib.pqn_bigmlp_gemm(m, n, k, a.data_ptr(), k, 0, b.data_ptr(), k, 0, None, c.data_ptr(), n, nsplit, m * n, tile,
                                   scratch.data_ptr(), _lib.stream_ptr()), "gemm")
torch.cuda.synchronize()
ref = a.double() @ b.double().T
print("max err", float((c.double().sum(0) - ref).abs().max()))

if os.environ.get("PQN_BM_STAMPS"):
    import ctypes
    buf = (ctypes.c_ulonglong * 128)()
    _lib.check(lib.pqn_debug_bm_stamps(buf), "stamps")
    t = list(buf)
    print("total K loop", t[2] - t[0], "first gload issue", t[1] - t[0])
    for s_ in range(0, 12):
        b = 8 + 8 * s_
        if t[b + 5] == 0: break
        print(f"step {s_}: wait+lstore {t[b+1]-t[b]:6d}  gload issue {t[b+2]-t[b+1]:6d}  frag wait+mfma {t[b+3]-t[b+2]:6d}  barrier {t[b+4]-t[b+3]:6d}  frag issue {t[b+5]-t[b+4]:6d}  | step total {t[b+5]-t[b]:6d}")
