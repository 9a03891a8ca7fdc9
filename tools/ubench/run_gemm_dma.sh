cd tools/ubench
for s in 1 2 3 4; do ./gemm_dma 1024 1024 1024 $s; done
for s in 1 2; do ./gemm_dma 2048 1024 1024 $s; done
for s in 1 2 3; do ./gemm_dma 1024 1024 1376 $s; done
for s in 2 4 8; do ./gemm_dma 1024 1024 2048 $s; done
