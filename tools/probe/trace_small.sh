# kernel timeline of a search of $1 queries (default 1) on the diagnostic build; extra environment as further arguments
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_small
mkdir -p $OUT
B=${1:-1}
export SWEEP=$B,$B LB_GPU_SO=$ROOT/longbow_amd/liblongbow_gpu_diag.so
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- python3 $ROOT/tools/bench_sweep.py > $OUT/run.log 2>&1
cd $ROOT
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-10:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  q{r.get('Queue_Id','?'):>3s}  {r['Kernel_Name'][:80]}")
PY
grep "B=" $OUT/run.log
