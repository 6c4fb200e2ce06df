# Kernel-by-kernel trace of the last search of a tools/bench_sweep.py run (rocprofv3 kernel trace; the profiler serialises the
# kernels, so the gaps are its own).  usage: BS="1 8" bash tools/probe/trace_search.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for B in ${BS:-1 8}; do
  rm -rf $ROOT/gpurun_out/tr_$B
  SWEEP=$B,$B rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/tr_$B -o t -- python3 $ROOT/tools/bench_sweep.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("$ROOT/gpurun_out/tr_$B/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # last search: from the last query_prep to the end
    # the last search: from the last launch that starts one (query preparation, or the sample it rides in up to 8 queries)
    idx = [i for i, r in enumerate(rows) if "query_prep" in r["Kernel_Name"] or "sample_scores_kernel" in r["Kernel_Name"]]
    prev = None
    for r in rows[idx[-1]:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("B=$B", r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-58:], "dur", round((e - s) / 1e3, 1), "gap", round((s - prev) / 1e3, 1) if prev else 0, "grid", r.get("Grid_Size", ""), "wg", r.get("Workgroup_Size", ""))
        prev = e
PY
done
