#!/bin/bash
# Regenerates the rocprofv3 evidence behind bench.py's roofline numbers.  Run ON the GPU box from the repo
# root:  bash tools/make_profiles.sh r01   ->  gpurun_out/prof_r01/{stats,fetch,write,sq}/...
# then   python tools/summarize_profiles.py r01   (anywhere) copies the summaries into profiles/.
# Counters are collected in their own passes (one --pmc group per run, kernel-trace only), as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -e
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[profiles] kernel stats"; 
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
echo "[profiles] FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
echo "[profiles] WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
echo "[profiles] SQ busy / LDS conflicts"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast > $OUT/sq.log 2>&1 || echo "[profiles] SQ pass failed (counter set not available); see $OUT/sq.log"
echo "[profiles] done"
