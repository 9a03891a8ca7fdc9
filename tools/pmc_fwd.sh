#!/bin/bash
# PMC counters of the forward kernel (separate passes; no trace domains combined with --pmc)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TARGET=${1:-tools/ablate_fwd.py}
KERN=${2:-qnet_cnn_fwd}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc
  rocprofv3 --pmc $set -d /tmp/pmc -o x -- python $R/$TARGET >/dev/null 2>&1
  python - "$KERN" <<'PY'
import sqlite3, sys, glob
db = sqlite3.connect(glob.glob('/tmp/pmc/*results.db')[0])
cur = db.cursor()
try:
    rows = cur.execute("select name, counter_name, avg(counter_value) from pmc_events where name like ? group by name, counter_name", ('%'+sys.argv[1]+'%',)).fetchall()
except Exception as e:
    cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
    print("pmc_events cols", cols); rows = []
for kname, cname, v in rows:
    print("%-32s %-28s %14.1f" % (kname[:32], cname, v))
PY
done
