#!/bin/bash
# HBM traffic of the dominant kernel (qnet_cnn_train_kernel / its pair form in bf16x3 mode) from PMC counters, separate passes,
# per MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE in KB; FETCH_SIZE under-reports wide coalesced reads by 2x.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  PQN_MODE=${PQN_MODE:-0} rocprofv3 --pmc $c -d /tmp/pmc_$c -o x -- python $R/tools/ablate_train.py >/dev/null 2>&1
done
python - <<'PY'
import sqlite3, glob, json, os
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(glob.glob(f'/tmp/pmc_{c}/*results.db')[0])
    for kern in ("qnet_cnn_train_", "qnet_fc1_wgrad", "qnet_grad_reduce_kernel", "radam_apply_kernel"):
        v = db.execute("select avg(counter_value), count(*) from pmc_events where name like ? and counter_name = ?", ('%'+kern+'%', c)).fetchone()
        out.setdefault(kern, {})[c + "_KB_avg"] = v[0]
        out[kern]["launches"] = v[1]
k = out["qnet_cnn_train_"]
mode = {"0": "f32", "1": "f16", "2": "bf16x3"}[os.environ.get("PQN_MODE", "0")]
res = {"kernel": "qnet_cnn_train_pair_kernel<4> (two tiles per workgroup)" if mode == "bf16x3" and os.environ.get("PQN_T1_PAIR", "1") != "0" else "qnet_cnn_train_kernel<4>", "workload": "4096-sample minibatch gathered from 32768 Breakout transitions",
       "matmul": mode, "seeds_per_launch": 1,
       "FETCH_SIZE_KB_avg": k["FETCH_SIZE_KB_avg"], "WRITE_SIZE_KB_avg": k["WRITE_SIZE_KB_avg"],
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated, taken as reported",
       "hbm_bytes_per_launch": (2 * k["FETCH_SIZE_KB_avg"] + k["WRITE_SIZE_KB_avg"]) * 1024.0,
       "all_kernels": out}
os.makedirs(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", f"pmc_train_kernel_{mode}.json"), "w"), indent=1)
print(json.dumps(res)[:1500])
PY
