#!/bin/bash
# HBM traffic of the kernels of the HEADLINE launch shape (16 seeds x 4096-sample minibatches per launch) from PMC
# counters: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; no trace domains) over a short run of bench.py,
# corrected as MI355X_MICROARCH.md "HBM" prescribes (KB units; FETCH_SIZE tallies wide coalesced reads at half size).
# Writes gpurun_out/pmc_train_kernel_<mode>_seeds16.json (copy to profiles/r03_...): bench.py's roofline.traffic reads it.
MODE=${1:-bf16x3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum; do
  rm -rf /tmp/pmcb_$c
  timeout 900 rocprofv3 --pmc $c -d /tmp/pmcb_$c -o x -- python $R/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --matmul-dtype $MODE >/dev/null 2>&1
done
MODE=$MODE python - <<'PY'
import sqlite3, glob, json, os
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCP_TCC_READ_REQ_sum"):
    db = sqlite3.connect(glob.glob(f'/tmp/pmcb_{c}/*results.db')[0])
    for kern in ("qnet_cnn_train_", "qnet_cnn_rollout_", "qnet_fc1_wgrad", "qnet_grad_reduce_kernel", "radam_apply_kernel"):
        v = db.execute("select avg(counter_value), count(*), min(name) from pmc_events where name like ? and counter_name = ?", ('%' + kern + '%', c)).fetchone()
        out.setdefault(kern, {})[c + ("_KB_avg" if c.endswith("SIZE") else "_avg")] = v[0]
        out[kern]["launches"] = v[1]
        out[kern]["name"] = (v[2] or "")[:60]
mode = os.environ["MODE"]
k = out["qnet_cnn_train_"]
res = {"kernel": k["name"], "workload": "bench.py headline: 16 seeds x (4096-sample minibatch gathered from 4096 envs x 32 steps of Breakout) per launch",
       "matmul": mode, "seeds_per_launch": 16,
       "FETCH_SIZE_KB_avg": k["FETCH_SIZE_KB_avg"], "WRITE_SIZE_KB_avg": k["WRITE_SIZE_KB_avg"],
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated, taken as reported",
       "hbm_bytes_per_launch": (2 * k["FETCH_SIZE_KB_avg"] + k["WRITE_SIZE_KB_avg"]) * 1024.0,
       "l2_to_cu_bytes_per_launch": (k.get("TCP_TCC_READ_REQ_sum_avg") or 0) * 64.0 or None,
       "l2_to_cu_note": "TCP_TCC_READ_REQ_sum (vector-L1 -> L2 read requests) x 64 B per request",
       "all_kernels": out}
dst = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
os.makedirs(dst, exist_ok=True)
json.dump(res, open(os.path.join(dst, f"pmc_train_kernel_{mode}_seeds16.json"), "w"), indent=1)
print(json.dumps(res)[:1800])
PY
