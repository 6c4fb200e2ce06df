#!/usr/bin/env python3
"""Kernel timeline of the last search in a rocprofv3 --kernel-trace CSV: start offset, duration and the gap to the
previous kernel.  usage: python tools/trace_timeline.py <..._kernel_trace.csv> [kernels per search]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:6.1f} us  {r['Kernel_Name'][:90]}")
    prev_end = e
print(f"span {(prev_end - t0) / 1e3:.1f} us")
