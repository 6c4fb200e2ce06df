set -x
cd $GRAFT_REPO_ROOT
bash tools/make_profiles.sh r03 > gpurun_out/make_profiles_r03.log 2>&1
echo "make_profiles rc=$?" >> gpurun_out/progress_final.txt
export LB_GPU_SO=$GRAFT_REPO_ROOT/longbow_amd/liblongbow_gpu_diag.so
bash tools/prof_tall.sh 4 > gpurun_out/prof_tall4.log 2>&1
echo "prof_tall 4 rc=$?" >> gpurun_out/progress_final.txt
bash tools/prof_tall.sh 2 > gpurun_out/prof_tall2.log 2>&1
echo "prof_tall 2 rc=$?" >> gpurun_out/progress_final.txt
bash tools/prof_mempath.sh 4 1024 > gpurun_out/mempath.log 2>&1
echo "mempath rc=$?" >> gpurun_out/progress_final.txt
timeout -k 10 500 python tools/route_grid.py > gpurun_out/route_grid_final.txt 2>&1
echo "route grid rc=$?" >> gpurun_out/progress_final.txt
unset LB_GPU_SO
timeout -k 10 600 python bench.py > gpurun_out/r3_bench_final.log 2> gpurun_out/r3_bench_final.err
echo "bench rc=$?" >> gpurun_out/progress_final.txt
