#!/bin/bash
# usage (on the GPU box): tools/prof_kernels.sh  -> per-kernel GPU times of the fused CNN kernels (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pf /tmp/pt
for a in ${FWD_ABLATES:-0}; do
rm -rf /tmp/pf
PQN_ABLATE=$a rocprofv3 --kernel-trace -d /tmp/pf -o x -- python $R/tools/ablate_fwd.py 4096 >/dev/null 2>&1
echo -n "fwd ablate=$a: "; python $R/tools/rocprof_summary.py /tmp/pf/x_results.db 3 | grep -E "qnet_cnn_fwd" | cut -c1-60
done
rocprofv3 --kernel-trace -d /tmp/pt -o x -- python $R/tools/ablate_train.py >/dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/pt/x_results.db 8 | grep -E "qnet|radam"
for a in ${TRAIN_ABLATES:-}; do
rm -rf /tmp/pt
PQN_ABLATE_TRAIN=$a rocprofv3 --kernel-trace -d /tmp/pt -o x -- python $R/tools/ablate_train.py >/dev/null 2>&1
echo -n "train ablate=$a: "; python $R/tools/rocprof_summary.py /tmp/pt/x_results.db 8 | grep -E "qnet_cnn_train" | cut -c1-60
done
