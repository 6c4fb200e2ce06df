#!/bin/bash
# The one-tile kernel's pass with parts of its epilogue switched off (diagnostic build, timing only: LB_F16_ABL 6 = no epilogue,
# 7 = no flush, 8 = nothing admitted; the searches are redone on other routes, so the kernel's own average is read from a
# rocprofv3 kernel trace).  usage: bash tools/probe/pass_abl.sh [B]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
B=${1:-128}
export LB_GPU_SO=$ROOT/longbow_amd/liblongbow_gpu_diag.so SWEEP=$B,$B,$B
cd /tmp && export TMPDIR=/tmp
for abl in 0 8 7 6 0; do
  rm -rf $ROOT/gpurun_out/pass_abl_$abl
  LB_F16_ABL=$abl rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/pass_abl_$abl -o t -- python3 $ROOT/tools/bench_sweep.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("$ROOT/gpurun_out/pass_abl_$abl/**/*kernel_trace.csv", recursive=True):
    d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f))
          if "narrow16p_kernel" in r["Kernel_Name"] and "false, false, true>" in r["Kernel_Name"].replace("(bool)", "") ]
    d = [x for x in d if x > 150]
    if d: print("B=$B LB_F16_ABL=$abl  one-tile pass (thresholds inside) launches", len(d), "median us", sorted(d)[len(d)//2], "min", min(d))
PY
done
