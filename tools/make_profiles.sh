#!/bin/bash
# Regenerates the rocprofv3 evidence behind bench.py's roofline numbers.  Run ON the GPU box from the repo
# root:  bash tools/make_profiles.sh r02   ->  gpurun_out/prof_r02/{stats,fetch,write,sq,pq_stats,pq_sq1,pq_sq2,pq_fetch}/...
# then   python tools/summarize_profiles.py r02   (anywhere) copies the summaries into profiles/.
# Counters are collected in their own passes (one --pmc group per run, kernel-trace only), as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes.  The program itself follows `--` (python3 <script>).
set -e
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[profiles] kernel stats (whole bench line, all legs)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
echo "[profiles] FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-legs > $OUT/fetch.log 2>&1
echo "[profiles] WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-legs > $OUT/write.log 2>&1
echo "[profiles] SQ busy / LDS conflicts"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast --no-legs > $OUT/sq.log 2>&1 || echo "[profiles] SQ pass failed (counter set not available); see $OUT/sq.log"
echo "[profiles] PQ / ADC (config 4): kernel stats, SQ counters, FETCH_SIZE"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pq_stats -o pq -- python3 $ROOT/tools/bench_pq.py 100 1 > $OUT/pq_stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pq_sq1 -o pq -- python3 $ROOT/tools/bench_pq.py 100 1 > $OUT/pq_sq1.log 2>&1 || echo "[profiles] pq_sq1 failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pq_sq2 -o pq -- python3 $ROOT/tools/bench_pq.py 100 1 > $OUT/pq_sq2.log 2>&1 || echo "[profiles] pq_sq2 failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pq_fetch -o pq -- python3 $ROOT/tools/bench_pq.py 100 1 > $OUT/pq_fetch.log 2>&1 || echo "[profiles] pq_fetch failed"
echo "[profiles] done"
