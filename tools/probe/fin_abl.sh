export LB_GPU_SO=$PWD/longbow_amd/liblongbow_gpu_diag.so
for abl in 0 1 2 3; do for g in 0 8 64; do echo "== abl $abl G $g"; LB_FINISH_ABL=$abl LB_FINISH_G=$g SWEEP=1 python3 tools/bench_sweep.py 2>&1 | grep "B="; done; done
