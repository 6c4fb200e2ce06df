# thresholds inside the candidate launch (TAUIN) on / off (diagnostic build: LB_TAUIN), same box, alternating
export LB_GPU_SO=$PWD/longbow_amd/liblongbow_gpu_diag.so
for v in "LB_TAUIN=1" "LB_TAUIN=0" "LB_TAUIN=1" "LB_TAUIN=0"; do echo "== $v"; env $v SWEEP=${SWEEP:-1,8,32,128} python3 tools/bench_sweep.py 2>&1 | grep "B=" | cut -c1-105; done
