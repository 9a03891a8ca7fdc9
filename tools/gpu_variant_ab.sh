#!/bin/bash
# A/B of prebuilt library variants (purejaxql_amd/csrc/variants/libpqn_hip_<name>.so, built here with extra -D flags) on
# the headline bench, plus L2 hit / miss counters of the default build.  Output under gpurun_out/ab/.
mkdir -p gpurun_out/ab
L=purejaxql_amd/csrc/libpqn_hip.so
cp $L /tmp/libpqn_default.so
run() {
  timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/ab/bench_$1.json 2> gpurun_out/ab/bench_$1.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab/bench_$1.json").read().strip().splitlines()[-1])
print("$1: value %.4g  ms/step %.2f  T1 us %.1f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
  if [ -n "$PROF" ]; then   # per-kernel averages of this variant
    (R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pv; rocprofv3 --kernel-trace -d /tmp/pv -o x -- python $R/bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1; python $R/tools/rocprof_summary.py /tmp/pv/x_results.db 5 | cut -c1-110 | tail -5)
  fi
}
run default
for v in "$@"; do
  cp purejaxql_amd/csrc/variants/libpqn_hip_$v.so $L
  run $v
done
cp /tmp/libpqn_default.so $L
run default2
if [ -z "$PMC_L2" ]; then exit 0; fi
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_l2
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/pmc_l2 -o x -- python $R/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/pmc_l2/*results.db')[0])
for kern in ("qnet_cnn_train_", "qnet_cnn_rollout_", "qnet_fc1_wgrad", "qnet_grad_reduce_kernel", "radam_apply_kernel"):
    r = {}
    for c in ("TCC_HIT_sum", "TCC_MISS_sum"):
        v = db.execute("select avg(counter_value), count(*) from pmc_events where name like ? and counter_name = ?", ('%' + kern + '%', c)).fetchone()
        r[c] = v[0]
    if r["TCC_HIT_sum"] is not None:
        print("%-26s L2 hit %.3g miss %.3g  hit rate %.3f" % (kern, r["TCC_HIT_sum"], r["TCC_MISS_sum"], r["TCC_HIT_sum"] / max(1.0, r["TCC_HIT_sum"] + r["TCC_MISS_sum"])))
PY
