# a batch that ends 65 .. 128 queries into a 256-query tile: the tail on the one-tile kernel (128) or one more 256-wide tile
export LB_GPU_SO=$PWD/longbow_amd/liblongbow_gpu_diag.so
for v in 128 64 128 64; do echo "== LB_F16_SPLIT_TAIL_MAX=$v"; LB_F16_SPLIT_TAIL_MAX=$v SWEEP=${SWEEP:-352,384,640,896} python3 tools/bench_sweep.py 2>&1 | grep "B=" | cut -c1-170; done
