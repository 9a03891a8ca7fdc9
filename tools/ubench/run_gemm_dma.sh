cd tools/ubench
./gemm_dma 1024 1024 1024 3
./gemm_dma 1024 1024 1024 2
./gemm_dma 1024 1024 1024 4
./gemm_dma 2048 1024 1024 2
./gemm_dma 1024 1024 1376 3
./gemm_dma 1024 1024 2048 3
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pg; rocprofv3 --kernel-trace -d /tmp/pg -o x -- python $GRAFT_REPO_ROOT/tools/bigmlp_gemm_bench.py 1024 1024 1024 64 3 200 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/pg/x_results.db 6 | cut -c1-150
rm -rf /tmp/pg; rocprofv3 --kernel-trace -d /tmp/pg -o x -- python $GRAFT_REPO_ROOT/tools/bigmlp_gemm_bench.py 2048 1024 1024 64 2 200 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/pg/x_results.db 4 | cut -c1-150
