# measured-residual error bound on / off (diagnostic build: LB_MEASURED_RHO), same box, alternating
export LB_GPU_SO=$PWD/longbow_amd/liblongbow_gpu_diag.so
for v in 1 0 1 0; do echo "== LB_MEASURED_RHO=$v"; LB_MEASURED_RHO=$v SWEEP=${SWEEP:-1,32,128,256,1024} python3 tools/bench_sweep.py 2>&1 | grep "B=" | cut -c1-190; done
for v in 1 0; do echo "== filtered LB_MEASURED_RHO=$v"; LB_MEASURED_RHO=$v SELS=10 BS=256 K=200 python3 tools/bench_filtered.py 2>&1 | tail -1 | cut -c1-200; done
