#!/bin/bash
# round 5: the position-parallel form (gather + cnn_pos_fwd_kernel + cnn_pos_bwd_kernel) -- parity tests, then the headline bench with
# the round-4 kernels (PQN_BWD_POS=0) and with the position form (1), each with its rocprofv3 kernel table.  (When the r05_v0 / r05_v1
# profiles were taken the script also ran the values 3 and 4 = round 2's backward / this backward behind the forward-only pair
# kernel; both paths were deleted later in the round.)
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_qnet_gpu.py -x -q -m gpu -k "position_parallel" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
run() {
  PQN_BWD_POS=$1 timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_pos$1.json 2> $O/bench_pos$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_pos$1.json").read().strip().splitlines()[-1])
    print("bwd_pos=$1: value %.4g  ms/step %.2f  T1 us %.1f  forms %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["config"].get("kernel_forms")))
except Exception as e:
    print("bwd_pos=$1 failed", e); print(open("$O/bench_pos$1.err").read()[-2000:])
PY
  (R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pv; PQN_BWD_POS=$1 timeout 600 rocprofv3 --kernel-trace -d /tmp/pv -o x -- python $R/bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1; python $R/tools/rocprof_summary.py /tmp/pv/x_results.db 8 | cut -c1-150 > $R/$O/kstats_pos$1.txt; tail -9 $R/$O/kstats_pos$1.txt)
}
run 0
run 1
PQN_BWD_POS=1 PQN_T1_STAMPS=1 timeout 300 python tools/pos_stamps.py > $O/stamps.txt 2>&1; tail -5 $O/stamps.txt
