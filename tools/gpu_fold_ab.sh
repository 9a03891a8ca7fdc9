#!/bin/bash
# round 6: fold + clip + RAdam in one launch (option fold_apply, pqn_fold.h) against the two launches, inside one gpurun call:
# the headline bench alternating, then the launch shapes below the position-parallel form's threshold
R=$PWD
O=$R/gpurun_out/foldab; mkdir -p $O
for rep in 1 2; do
  for f in 2 0; do
    PQN_FOLD_APPLY=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_f${f}_$rep.json 2>$O/err_f${f}_$rep.txt
    python - $O/bench_f${f}_$rep.json $f <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("fold_apply=%s  value %.4g  ms %.3f" % (sys.argv[2], d["value"], d["ms_per_step"]))
P
  done
done
for s in "128 1 100" "1024 1 60" "4096 1 30" "1024 2 60" "4096 2 30" "1024 4 40" "4096 4 20" "4096 8 16"; do
  set -- $s
  for f in 2 0; do echo -n "fold_apply=$f  "; PQN_FOLD_APPLY=$f timeout 200 python tools/shape_run.py $1 $2 $3 | tail -1; done
done
