"""S seeds of the bench workload batched into the same launches: aggregate rate (used under rocprofv3)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from purejaxql_amd import _lib
_lib.load()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = bench.workload_config(4096, "seeds")
cfg["TOTAL_TIMESTEPS"] = (steps + 6) * cfg["NUM_ENVS"] * cfg["NUM_STEPS"]
print(json.dumps(bench.multi_seed_rate(cfg, S, steps, 3, torch.device("cuda", 0))))
