# kernel timeline of a C4 search (tools/bench_pq.py 100 $1) on the release build
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_pq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- python3 $ROOT/tools/bench_pq.py 100 ${1:-1} > $OUT/run.log 2>&1
cd $ROOT
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-16:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us ({(e-s)/1e3:7.1f})  {r['Kernel_Name'][:110]}")
PY
grep -v "^[WE]2026" $OUT/run.log | tail -6
