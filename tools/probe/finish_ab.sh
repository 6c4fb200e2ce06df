#!/bin/bash
# Finish launch, tiled form (17+ queries), A/B on the diagnostic build: depth of the row ring
# (LB_FINISH_NST), depth of the row ring (LB_FINISH_NST), ring against register tile (LB_FINISH_RING_MAXQ); filtered config-5
# share and the unfiltered sweep.  Run ON the GPU box from the repo root: bash tools/probe/finish_ab.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/finish_ab
mkdir -p $OUT
export LB_GPU_SO=$ROOT/longbow_amd/liblongbow_gpu_diag.so
cd $ROOT
for nst in 4 2; do
  echo "== LB_FINISH_NST=$nst: filtered 10 / 50 / 90 % visible, K=200"; LB_FINISH_NST=$nst K=200 SELS=10,50,90 BS=32,256 python3 tools/bench_filtered.py 2>&1 | grep "^sel" | sed 's/route.*fallbacks/fallbacks/' | cut -c1-250 | tee $OUT/filt_nst$nst.txt
done
echo "== default: 1M x 768 sweep"; SWEEP=24,64,128,256,512,1024 python3 tools/bench_sweep.py 2>&1 | grep "^B=" | cut -c1-220 | tee $OUT/sweep.txt
echo "== ring at every batch size (LB_FINISH_RING_MAXQ=4096)"; LB_FINISH_RING_MAXQ=4096 SWEEP=512,1024 python3 tools/bench_sweep.py 2>&1 | grep "^B=" | cut -c1-220 | tee $OUT/sweep_ring.txt
