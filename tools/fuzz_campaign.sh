#!/bin/bash
# Randomised parity campaign on the GPU box (fresh seeds on every call: pass a base seed).  Every tool compares the
# HIP path with the oracle or with the strict mode (itself oracle-checked in tests/); a mismatch makes the tool exit 1.
# usage (from the repo root, on the GPU box): bash tools/fuzz_campaign.sh [base_seed=21] [seconds_per_sweep=170]
#   -> gpurun_out/campaign/*.log and gpurun_out/campaign/summary.txt (copy the summary into profiles/)
S=${1:-21}
T=${2:-170}
OUT=gpurun_out/campaign
mkdir -p $OUT
rc=0
run() { # name, command...
    local name=$1; shift
    echo "[campaign] $name" | tee -a $OUT/summary.txt
    if timeout -k 10 $((T + 240)) "$@" > $OUT/$name.log 2>&1; then
        echo "    ok: $(grep -E 'cases, [0-9]+ mismatches|passed|mismatch|OK|ok' $OUT/$name.log | tail -1)" | tee -a $OUT/summary.txt
    else
        echo "    FAILED (exit $?): $(tail -2 $OUT/$name.log | tr '\n' ' ')" | tee -a $OUT/summary.txt
        rc=1
    fi
}
: > $OUT/summary.txt
echo "# tools/fuzz_campaign.sh $S $T on $(date -u +%FT%TZ), $(python3 -c 'import torch;print(torch.cuda.get_device_name(0))' 2>/dev/null)" >> $OUT/summary.txt
run fuzz_copy_seed$S python3 tools/probe/fuzz_copy.py $S $T
run fuzz_copy_seed$((S + 1)) python3 tools/probe/fuzz_copy.py $((S + 1)) $T
FUZZ_WIDE=1 run fuzz_copy_wide_seed$((S + 2)) python3 tools/probe/fuzz_copy.py $((S + 2)) $T
run fuzz_scales_seed$((S + 3)) python3 tools/probe/fuzz_scales.py $((S + 3)) $T
for r in $((S)) $((S + 1)) $((S + 2)); do
    LB_FUZZ_ROUND=$r run pytest_fuzz_round$r python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -x
done
run stress python3 tools/stress.py
run stress_fused python3 tools/stress_fused.py
run robustness python3 tools/robustness.py
echo "[campaign] exit $rc" | tee -a $OUT/summary.txt
exit $rc
