#!/bin/bash
# Memory-path counters (TA / TCP / UTCL1 / TCC / TD) of a candidate kernel at 1M x 768 (run ON the GPU box from the repo root).
# usage: bash tools/prof_mempath.sh [CAND_MODE] [B]  -> gpurun_out/prof_mem_m<CAND_MODE>_b<B>/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
M=${1:-4}; B=${2:-1024}
OUT=$ROOT/gpurun_out/prof_mem_m${M}_b${B}
mkdir -p $OUT
export CAND_MODE=$M SWEEP=$B,$B
cd /tmp && export TMPDIR=/tmp
# (at most 2 TA, 4 TCP, 4 TCC, 2 TD counters per pass: more and rocprofv3 aborts with "exceeds the capabilities of the hardware" and then hangs)
run() { n=$1; shift; timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o t -- python3 $ROOT/tools/bench_sweep.py > $OUT/$n.log 2>&1 || echo "pass $n failed"; echo "pass $n done" >> $OUT/progress.txt; }
run m1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
run m2 TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
run m3 TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum
run m4 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum
cd $ROOT
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
out = sys.argv[1]
for d in ("m1", "m2", "m3", "m4"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
            if "gemm_filter" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for k, v in acc.items():
            print(d, k, "launches", len(dur[k]), "total ns", sum(dur[k].values()))
            for c, x in sorted(v.items()): print(f"    {c:40s} {x:.4g}")
PY
