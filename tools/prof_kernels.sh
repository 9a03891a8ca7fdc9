#!/bin/bash
# usage (on the GPU box): tools/prof_kernels.sh  -> per-kernel GPU times of the fused CNN kernels (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pf /tmp/pt
rocprofv3 --kernel-trace -d /tmp/pf -o x -- python $R/tools/ablate_fwd.py 4096 >/dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/pf/x_results.db 3 | grep -E "qnet_cnn_fwd"
rocprofv3 --kernel-trace -d /tmp/pt -o x -- python $R/tools/ablate_train.py >/dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/pt/x_results.db 8 | grep -E "qnet|radam"
