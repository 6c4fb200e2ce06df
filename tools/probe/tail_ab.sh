# A/B: a batch that ends in a partial 256-query tile, tail on the one-tile kernel (1) or in a padded 256-wide tile (0); diagnostic build
export LB_GPU_SO=$PWD/longbow_amd/liblongbow_gpu_diag.so
for w in 0 1; do
  echo "== LB_F16_SPLIT_TAIL=$w"
  LB_F16_SPLIT_TAIL=$w SWEEP=${SWEEP:-257,320,384,512,640,896,1024} timeout -k 10 200 python tools/bench_sweep.py 2>&1 | grep "B=" || exit 1
done
