"""Per-step cycle stamps of the bf16x3 fc1 weight-gradient kernel (T2), workgroup 0 / wave 0, taken while the headline
bench configuration runs in this process (PQN_T1_STAMPS=1 must be set before the library loads)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("PQN_T1_STAMPS", "1")
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-extras", "--matmul-dtype", "bf16x3"] + sys.argv[1:]
import bench
bench.main()
from purejaxql_amd import _lib
buf = (ctypes.c_ulonglong * 32)()
lib = _lib.load()
lib.pqn_debug_t2_stamps.argtypes = [ctypes.c_void_p]
rc = lib.pqn_debug_t2_stamps(buf)
s = list(buf)
print("rc", rc, "slab resident after %d cycles; steps of row block 0: %s; row blocks: %s; total %d" % (
    s[1] - s[0], [s[2 + u] - (s[1] if u == 0 else s[1 + u]) for u in range(8)],
    [s[10 + k] - (s[9] if k == 0 else s[9 + k]) for k in range(16) if s[10 + k]], max(s) - s[0]))
