#!/bin/bash
# round 4, call A: seed-group pipeline (pqn_cnn_update_seed_groups) -- parity, A/B of the tail placements, kernel trace,
# and the phase stamps of the single-tile T1 form in all three operand modes (+ the pair form)
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_headline_gpu.py -x -q -k "seed_groups or 16_seeds" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() {
  name=$1; shift
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name: value %.4g  ms/step %.2f  T1 us %.1f  driver %s groups %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["config"]["driver"], d["config"].get("seed_groups")))
except Exception as e:
    print("$name: FAILED", repr(e)); print(open("$O/bench_$name.err").read()[-1500:])
PY
}
run g1 --seed-groups 1
run g2 --seed-groups 2
run g2e --seed-groups 2 --groups-tail eager
run g2m224 --seed-groups 2 --groups-tail masked:224:256
run g2mod8 --seed-groups 2 --groups-tail maskmod:8:7
run g4 --seed-groups 4
run g1b --seed-groups 1
run g2b --seed-groups 2
R=$PWD
(cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pv; timeout 600 rocprofv3 --kernel-trace -d /tmp/pv -o x -- python $R/bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline --seed-groups 2 > /dev/null 2>&1; python $R/tools/rocprof_summary.py /tmp/pv/x_results.db 12 | cut -c1-150) > $O/kernel_stats_g2.txt 2>&1
tail -14 $O/kernel_stats_g2.txt
for m in 0 1 2; do
  PQN_T1_STAMPS=1 PQN_MODE=$m PQN_T1_PAIR=0 timeout 300 python tools/ablate_train.py > $O/stamps_single_mode$m.txt 2>&1; tail -5 $O/stamps_single_mode$m.txt
done
PQN_T1_STAMPS=1 PQN_SEED_GROUPS=1 timeout 300 python tools/bench_stamps.py > $O/stamps_pair_16seeds.txt 2>&1; tail -4 $O/stamps_pair_16seeds.txt
