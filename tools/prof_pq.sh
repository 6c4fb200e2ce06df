#!/bin/bash
# rocprofv3 evidence for the PQ/ADC search kernels (BASELINE config 4).  Run ON the GPU box from the repo root:
#   bash tools/prof_pq.sh r02   ->  gpurun_out/prof_r02/{pq_stats,pq_sq1,pq_sq2}
# Counters are collected in their own passes (kernel-trace only), per MI355X_MICROARCH.md.
set -e
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pq_stats -o pq -- python3 $ROOT/tools/bench_pq.py 100 1 > $OUT/pq_stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pq_sq1 -o pq -- python3 $ROOT/tools/bench_pq.py 100 1 > $OUT/pq_sq1.log 2>&1 || echo "sq1 pass failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pq_sq2 -o pq -- python3 $ROOT/tools/bench_pq.py 100 1 > $OUT/pq_sq2.log 2>&1 || echo "sq2 pass failed"
echo done
