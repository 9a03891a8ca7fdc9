#!/bin/bash
# L2 hit / miss of the wide-MLP GEMM kernel
mkdir -p gpurun_out/r3g
O=$PWD/gpurun_out/r3g
R=$PWD; cd /tmp && export TMPDIR=/tmp
for cfg in "2048 1024 1024 128 3" "2048 1024 1024 128 1" "1024 1024 1024 64 3"; do
  for ctr in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE"; do
    rm -rf /tmp/pg
    timeout 200 rocprofv3 --pmc $ctr -d /tmp/pg -o x -- python $R/tools/bigmlp_gemm_bench.py $cfg 10 > /tmp/pg.log 2>&1
    python - <<PY
import sqlite3, glob
try:
    db = sqlite3.connect(glob.glob('/tmp/pg/*results.db')[0])
    rows = db.execute("select counter_name, avg(counter_value) from pmc_events where name like '%bm_gemm%' group by counter_name").fetchall()
    print("$cfg |", "  ".join(f"{n} {v:.4g}" for n, v in rows))
except Exception as e:
    print("$cfg | $ctr failed", repr(e)[:100])
PY
  done
done | tee $O/gemm_pmc.txt
