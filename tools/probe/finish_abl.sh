#!/bin/bash
# Where the finish launch's time goes at 256 queries (filtered config-5 share and unfiltered 1M x 768): timing-only ablations
# of the diagnostic build (LB_FINISH_ABL: 1 = no gather / exact sums, 2 = no radix select, 3 = return at once).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export LB_GPU_SO=$ROOT/longbow_amd/liblongbow_gpu_diag.so
cd $ROOT
for abl in 0 1 2 3; do
  echo "== LB_FINISH_ABL=$abl filtered 10 % B=256 K=200"; LB_FINISH_ABL=$abl K=200 SELS=10 BS=256 python3 tools/bench_filtered.py 2>&1 | grep "^sel" | sed 's/route.*fallbacks/fallbacks/' | cut -c1-260
  echo "== LB_FINISH_ABL=$abl unfiltered 1M x 768 B=64,256,1024"; LB_FINISH_ABL=$abl SWEEP=64,256,1024 python3 tools/bench_sweep.py 2>&1 | grep "^B=" | cut -c1-220
done
