#!/bin/bash
# Filtered search (config-5 share), A/B on the diagnostic build: cache policy of the row-list gather, epilogue / flush
# ablations, kernel trace.  Run ON the GPU box from the repo root: bash tools/probe/filt_ab.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/filt_ab
mkdir -p $OUT
export LB_GPU_SO=$ROOT/longbow_amd/liblongbow_gpu_diag.so K=200
cd $ROOT
echo "== default"; SELS=${SELS:-10,50} BS=${BS:-32,256} python3 tools/bench_filtered.py | tee $OUT/default.txt
echo "== LB_F16_MAPPED_NT=1"; LB_F16_MAPPED_NT=1 SELS=${SELS:-10,50} BS=${BS:-32,256} python3 tools/bench_filtered.py | tee $OUT/nt.txt
echo "== no epilogue (timing only)"; LB_F16_ABL=6 SELS=10 BS=256 python3 tools/bench_filtered.py | tee $OUT/abl6.txt
echo "== no flush (timing only)"; LB_F16_ABL=7 SELS=10 BS=256 python3 tools/bench_filtered.py | tee $OUT/abl7.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o f -- python3 $ROOT/tools/bench_filtered.py > $OUT/trace.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, os
for f in glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/filt_ab/trace/**/*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:25]:
        print(r["Name"].replace("(anonymous namespace)::", "")[:110], r["Calls"], r["AverageNs"], r["Percentage"])
PY
