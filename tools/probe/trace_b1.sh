# kernel timeline of a single-query search over the fp16 copy (1M x 768 cosine, k = 100): rocprofv3 --kernel-trace, last search
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_b${1:-1}
mkdir -p $OUT
export SWEEP=${1:-1},${1:-1}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- python3 $ROOT/tools/bench_sweep.py > $OUT/run.log 2>&1
cd $ROOT
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f ${2:-7} | tee $OUT/timeline.txt
grep "B=" $OUT/run.log
