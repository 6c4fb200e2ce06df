#!/bin/bash
# SQ counters for the split-bf16 candidate kernels at 1M x 768, 1024 queries (run ON the GPU box from the repo root).
# usage: bash tools/prof_tall.sh [CAND_MODE]   -> gpurun_out/prof_tall_m<CAND_MODE>/<pass>/...   (4 = fp16 single product, 2 = split in registers, 1 = image)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_tall_m${1:-1}${2:+_b$2}
mkdir -p $OUT
export CAND_MODE=${1:-1} SWEEP=${2:-1024},${2:-1024}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq1 -o t -- python3 $ROOT/tools/bench_sweep.py > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/sq2 -o t -- python3 $ROOT/tools/bench_sweep.py > $OUT/sq2.log 2>&1
cd $ROOT
export PT_OUT=$OUT
python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections
print("rocprofv3 --pmc (two passes) of: CAND_MODE=%s SWEEP=%s tools/bench_sweep.py  (1M x 768 cosine, k = 100)" % (__import__("os").environ.get("CAND_MODE"), __import__("os").environ.get("SWEEP")))
tot = {}
for d in ("sq1", "sq2"):
    for f in glob.glob(f"{__import__('os').environ['PT_OUT']}/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
            if "gemm_filter" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for k, v in acc.items():
            print(d, k, "launches", len(dur[k]), "total ns", sum(dur[k].values()))
            for c, x in sorted(v.items()): print(f"    {c:28s} {x:.4g}")
            tot.setdefault(k, {}).update(v); tot[k]["ns_" + d] = sum(dur[k].values())
for k, v in tot.items():
    if "SQ_BUSY_CYCLES" in v and v.get("ns_sq1"):
        clk = v["SQ_BUSY_CYCLES"] / 32 / v["ns_sq1"]
        print(f"derived {k}: shader clock ~ SQ_BUSY_CYCLES/32/ns = {clk:.2f} GHz; MFMA pipe busy = MFMA_BUSY/(32 SQ_BUSY_CYCLES) = "
              f"{v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (32 * v['SQ_BUSY_CYCLES']):.3f}; LDS index unit busy = LDS_IDX_ACTIVE/(256 CUs * clock * ns) = "
              f"{v.get('SQ_LDS_IDX_ACTIVE', 0) / (256 * clk * v['ns_sq1']):.3f}; bank conflict cycles = {int(v.get('SQ_LDS_BANK_CONFLICT', 0))}")
PY
