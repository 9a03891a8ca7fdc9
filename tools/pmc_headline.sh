#!/bin/bash
# PMC survey of the headline kernels (T1 pair, T2, rollout pair, fold, RAdam): LDS conflicts, MFMA / VALU busy, waits, instruction mix.
# One rocprofv3 --pmc pass per counter pair over `bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline`; value = average of the
# counter over the kernel's rows.  SQ counters are sampled: use ratios.  Output: gpurun_out/pmc_headline.txt -> profiles/r0N_*_headline_pmc.txt
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_INSTS_VMEM" "SQ_INSTS_MFMA SQ_INSTS_SALU"; do
  rm -rf /tmp/pp
  timeout 300 rocprofv3 --pmc $ctr -d /tmp/pp -o x -- python $R/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /tmp/pp.log 2>&1
  CTR="$ctr" python - <<'PY'
import sqlite3, glob, os
try:
    db = sqlite3.connect(glob.glob('/tmp/pp/*results.db')[0])
    for kern in ("qnet_cnn_train_pair", "qnet_fc1_wgrad_x3", "qnet_cnn_rollout_pair", "qnet_grad_reduce", "radam_apply"):
        rows = db.execute("select counter_name, avg(counter_value) from pmc_events where name like ? group by counter_name", ('%' + kern + '%',)).fetchall()
        print("%-24s" % kern, "  ".join(f"{n} {v:.4g}" for n, v in rows))
except Exception as e:
    print(os.environ["CTR"], "failed", repr(e)[:200])
PY
done | tee $O/pmc_headline.txt
