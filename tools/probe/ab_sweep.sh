# A/B of two builds of the library on one box: the batch sweep with each (LB_GPU_SO selects the build)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for so in ${A_SO:-longbow_amd/liblongbow_gpu_prev.so} ${B_SO:-longbow_amd/liblongbow_gpu.so}; do
  for rep in 1 2; do
    echo "== $so (run $rep)"
    LB_GPU_SO=$ROOT/$so SWEEP=${SWEEP:-1,8,32,64,128} python3 $ROOT/tools/bench_sweep.py 2>&1 | grep "B="
  done
done
