#!/bin/bash
# PMC passes over the headline launch shape (bench.py --no-extras: 16 seeds x 4096-sample minibatches per launch) for the kernels of
# the position-parallel form.  (1) HBM traffic: separate rocprofv3 --pmc passes for FETCH_SIZE / WRITE_SIZE / TCP_TCC_READ_REQ_sum (no
# trace domains), corrected as MI355X_MICROARCH.md "HBM" prescribes (KB units; FETCH_SIZE tallies wide coalesced reads at half size)
# -> gpurun_out/pmc_pos_bwd_kernel_<MD>_seeds16.json (copy to profiles/r06_...: bench.py's roofline.traffic reads it).  MD = operand mode
# (env, default f16x2); COMMIT = the commit the library was built from (env, recorded in the json).
# (2) SQ survey (instruction mix, MFMA / VALU busy, waits, LDS conflicts; SQ counters are sampled: use ratios) -> gpurun_out/pmc_pos_sq.txt
R=$PWD; O=$R/gpurun_out; mkdir -p $O
MD=${MD:-f16x2}; export MD
cd /tmp && export TMPDIR=/tmp
KERNS="cnn_pos_bwd_kernel cnn_pos_fwd_kernel cnn_pos_rollout_kernel pos_gather_kernel qnet_grad_reduce_kernel radam_apply_kernel"
for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum; do
  rm -rf /tmp/pmcb_$c
  timeout 600 rocprofv3 --pmc $c -d /tmp/pmcb_$c -o x -- python $R/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --matmul-dtype $MD >/dev/null 2>&1
done
KERNS="$KERNS" OUT=$O python - <<'PY'
import sqlite3, glob, json, os, datetime
out = {}
md = os.environ.get("MD", "f16x2")
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCP_TCC_READ_REQ_sum"):
    db = sqlite3.connect(glob.glob(f'/tmp/pmcb_{c}/*results.db')[0])
    for kern in os.environ["KERNS"].split():
        v = db.execute("select avg(counter_value), count(*), min(name) from pmc_events where name like ? and counter_name = ?", ('%' + kern + '%', c)).fetchone()
        out.setdefault(kern, {})[c + ("_KB_avg" if c.endswith("SIZE") else "_avg")] = v[0]
        out[kern]["launches"] = v[1]
        out[kern]["name"] = (v[2] or "")[:60]
k = out["cnn_pos_bwd_kernel"]
hb = lambda d: (2 * (d.get("FETCH_SIZE_KB_avg") or 0) + (d.get("WRITE_SIZE_KB_avg") or 0)) * 1024.0
res = {"kernel": k["name"], "workload": "bench.py headline: 16 seeds x (4096-sample minibatch gathered from 4096 envs x 32 steps of Breakout) per launch",
       "matmul": md, "seeds_per_launch": 16, "date": datetime.date.today().isoformat(), "commit": os.environ.get("COMMIT", "unknown"),
       "FETCH_SIZE_KB_avg": k["FETCH_SIZE_KB_avg"], "WRITE_SIZE_KB_avg": k["WRITE_SIZE_KB_avg"],
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated, taken as reported",
       "hbm_bytes_per_launch": hb(k),
       "l2_to_cu_bytes_per_launch": (k.get("TCP_TCC_READ_REQ_sum_avg") or 0) * 64.0 or None,
       "l2_to_cu_note": "TCP_TCC_READ_REQ_sum (vector-L1 -> L2 read requests) x 64 B per request; the LDS-DMA of the dz planes is part of it",
       "hbm_bytes_per_launch_by_kernel": {kk: hb(v) for kk, v in out.items()},
       "all_kernels": out}
json.dump(res, open(os.path.join(os.environ["OUT"], f"pmc_pos_bwd_kernel_{md}_seeds16.json"), "w"), indent=1)
print(json.dumps(res)[:1500])
PY
for ctr in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_INSTS_VMEM" "SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_ANY"; do
  rm -rf /tmp/pp
  timeout 300 rocprofv3 --pmc $ctr -d /tmp/pp -o x -- python $R/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --matmul-dtype $MD > /tmp/pp.log 2>&1
  CTR="$ctr" KERNS="$KERNS" python - <<'PY'
import sqlite3, glob, os
try:
    db = sqlite3.connect(glob.glob('/tmp/pp/*results.db')[0])
    for kern in os.environ["KERNS"].split()[:3]:
        rows = db.execute("select counter_name, avg(counter_value) from pmc_events where name like ? group by counter_name", ('%' + kern + '%',)).fetchall()
        print("%-24s" % kern, "  ".join(f"{n} {v:.4g}" for n, v in rows))
except Exception as e:
    print(os.environ["CTR"], "failed", repr(e)[:200])
PY
done | tee $O/pmc_pos_sq.txt
