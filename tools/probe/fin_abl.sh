export LB_GPU_SO=$PWD/longbow_amd/liblongbow_gpu_diag.so
for abl in 0 16 32 48; do echo "== abl $abl"; LB_FINISH_ABL=$abl SWEEP=1,32,64,128 python3 tools/bench_sweep.py 2>&1 | grep "B="; done
echo "== G=1 at 32"; LB_FINISH_G=1 SWEEP=32 python3 tools/bench_sweep.py 2>&1 | grep "B="
