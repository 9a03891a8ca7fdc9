"""Cycle stamps of one iteration (super-tile 8) of the position-parallel backward kernel, workgroup 0: producer wave 0
and consumer wave 4 (PQN_BWD_POS=1 PQN_T1_STAMPS=1 must be in the environment before the library loads)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("PQN_T1_STAMPS", "1"); os.environ.setdefault("PQN_BWD_POS", "1")
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-extras"]
import bench
bench.main()
from purejaxql_amd import _lib
buf = (ctypes.c_ulonglong * 32)()
print("rc", _lib.load().pqn_debug_t2_stamps(buf))
s = list(buf)
f = s[0:7]; b = s[16:22]
print("producer: masks+conv issue %d | conv drain+LN fwd %d | dgrad issue %d | drain+LN bwd %d | split+exchange %d | barrier wait %d | total %d" % (
    f[1] - f[0], f[2] - f[1], f[3] - f[2], f[4] - f[3], f[5] - f[4], f[6] - f[5], f[6] - f[0]))
print("consumer: pf issue %d | dW1+cw pos0 + dW1 pos1 %d | cw pos1 %d | pf_store (waits for the loads) %d | barrier wait %d | total %d" % (
    b[1] - b[0], b[2] - b[1], b[3] - b[2], b[4] - b[3], b[5] - b[4], b[5] - b[0]))
