export LB_GPU_SO=$PWD/longbow_amd/liblongbow_gpu_diag.so
for B in 256 1024; do
 for a in 0 1 2 3; do echo "B=$B abl=$a"; LB_F16_ABL=$a CAND_MODE=4 SWEEP=$B,$B python tools/bench_sweep.py 2>&1 | grep "B=" | tail -1; done
 echo "B=$B stamps"; LB_F16_ABL=5 python tools/tall16_probe.py $B 2>&1 | grep -v amdgpu
done
