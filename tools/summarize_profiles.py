#!/usr/bin/env python3
"""Turn gpurun_out/prof_<round>/ (tools/make_profiles.sh) into the committed summaries under profiles/:
  <round>_kernel_stats_bench.csv   rocprofv3 --stats per-kernel table (top rows)
  <round>_pmc_hbm_traffic.csv      per-kernel FETCH_SIZE / WRITE_SIZE sums, FETCH corrected x2 (gfx950)
  <round>_pmc_sq_gemm.txt          SQ counters of the largest gemm_filter_kernel launch
  traffic.json                     what bench.py reads for roofline.traffic
usage: python tools/summarize_profiles.py r01"""
import csv, glob, json, os, sys
from collections import defaultdict

R = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
DST = os.path.join(ROOT, "profiles")


def find(sub, suffix):
    hits = sorted(glob.glob(os.path.join(SRC, sub, "**", f"*{suffix}"), recursive=True))
    return hits[-1] if hits else None


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()


# ---- kernel stats
f = find("stats", "kernel_stats.csv")
if f:
    rows = list(csv.reader(open(f)))
    with open(os.path.join(DST, f"{R}_kernel_stats_bench.csv"), "w", newline="") as o:
        csv.writer(o, quoting=csv.QUOTE_NONNUMERIC).writerows(rows[:40])
    print("kernel stats ->", f"{R}_kernel_stats_bench.csv")


def counter_sums(sub):
    """{kernel: {counter: [per-dispatch value,...]}} from a counter_collection.csv"""
    f = find(sub, "counter_collection.csv")
    out = defaultdict(lambda: defaultdict(dict))
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        d = int(r["Dispatch_Id"])
        out[k][r["Counter_Name"]][d] = out[k][r["Counter_Name"]].get(d, 0.0) + float(r["Counter_Value"])
    return out


fetch, write = counter_sums("fetch"), counter_sums("write")
traffic = {"source": f"profiles/{R}_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes; "
                     "FETCH corrected x2 per MI355X_MICROARCH.md)",
           "note": "per-launch average over a batch's gemm launches (like roofline.achieved); counts L2 misses incl. Infinity-Cache hits"}
lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of: bench.py --steps 2 --warmup 1 --no-cpu-baseline",
         "# FETCH_SIZE is in KB and counts 1/2 of the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM): corrected = KB*1024*2",
         "kernel,launches,fetch_bytes_corrected_per_launch,write_bytes_per_launch,largest_launch_fetch_bytes_corrected"]
for k in sorted(set(fetch) | set(write)):
    fv = list(fetch[k].get("FETCH_SIZE", {}).values())
    wv = list(write[k].get("WRITE_SIZE", {}).values())
    if not fv and not wv:
        continue
    n = max(len(fv), len(wv))
    fb = sum(fv) * 1024 * 2 / max(len(fv), 1)
    wb = sum(wv) * 1024 / max(len(wv), 1)
    big = max(fv) * 1024 * 2 if fv else 0
    lines.append(f"{k},{n},{fb:.0f},{wb:.0f},{big:.0f}")
    if k.startswith("lb::gemm_filter_kernel"):
        tag = "gemm_filter_kernel_split" if k.rstrip(">").endswith("true") else "gemm_filter_kernel"
        traffic[f"{tag}_hbm_bytes_per_launch"] = fb + wb
        traffic[f"{tag}_largest_launch_bytes"] = big
        traffic[f"{tag}_launches_profiled"] = n
    if k.startswith("lb::scan_kernel"):
        traffic["scan_kernel_hbm_bytes_per_launch"] = fb + wb
        traffic["scan_kernel_launches_profiled"] = n
if len(lines) > 3:
    open(os.path.join(DST, f"{R}_pmc_hbm_traffic.csv"), "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
    print("traffic ->", f"{R}_pmc_hbm_traffic.csv, traffic.json")

# ---- SQ counters of the largest gemm launch
f = find("sq", "counter_collection.csv")
if f:
    per = defaultdict(dict)
    meta = {}
    for r in csv.DictReader(open(f)):
        if "gemm_filter_kernel" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        meta[d] = (int(r.get("Grid_Size", 0) or 0), int(r.get("End_Timestamp", 0) or 0) - int(r.get("Start_Timestamp", 0) or 0))
    if per:
        d = max(per, key=lambda x: meta[x][0])
        c = per[d]
        grid, dur = meta[d]
        out = ["rocprofv3 --pmc " + " ".join(sorted(c)),
               "largest gemm_filter_kernel launch of a batch (bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast)"]
        out += [f"{k} = {int(v)}" for k, v in sorted(c.items())]
        out += [f"dur = {dur}", f"grid = {grid}"]
        if dur > 0 and "SQ_BUSY_CYCLES" in c:
            clk = c["SQ_BUSY_CYCLES"] / 32 / dur
            busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * clk * dur) if clk else 0
            out.append(f"derived: shader clock ~ SQ_BUSY_CYCLES/32/dur = {clk:.2f} GHz; MFMA pipe busy = MFMA_BUSY/(1024 SIMDs * clock * dur) = {busy:.3f}; "
                       f"LDS bank conflict cycles = {int(c.get('SQ_LDS_BANK_CONFLICT', 0))}")
        open(os.path.join(DST, f"{R}_pmc_sq_gemm.txt"), "w").write("\n".join(out) + "\n")
        print("sq ->", f"{R}_pmc_sq_gemm.txt")


# ---- PQ / ADC kernels (config 4): stats + SQ counters + traffic of the code-pass kernels
f = find("pq_stats", "kernel_stats.csv")
if f:
    rows = list(csv.reader(open(f)))
    with open(os.path.join(DST, f"{R}_kernel_stats_pq.csv"), "w", newline="") as o:
        csv.writer(o, quoting=csv.QUOTE_NONNUMERIC).writerows(rows[:20])
    print("pq kernel stats ->", f"{R}_kernel_stats_pq.csv")
out = ["# tools/bench_pq.py 100 1 (100M x m=96 codes, k=100, B=1) under rocprofv3 --pmc (kernel-trace only), one counter group per run.",
       "# Per-dispatch medians over the launches of each code-pass kernel; SQ_* are summed over the chip by rocprofv3.",
       "# adc_prefilter_kernel = byte-table pass (ds_read_u8 gathers); adc_scan_dma_kernel = exact f32-table pass (ds_read_b32 gathers)."]
for sub in ("pq_sq1", "pq_sq2", "pq_fetch"):
    f = find(sub, "counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(dict))
    dur = defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "adc_prefilter_kernel" not in k and "adc_scan_dma_kernel" not in k:
            continue
        d = int(r["Dispatch_Id"])
        acc[k][r["Counter_Name"]][d] = acc[k][r["Counter_Name"]].get(d, 0.0) + float(r["Counter_Value"])
        dur[k][d] = int(r.get("End_Timestamp", 0) or 0) - int(r.get("Start_Timestamp", 0) or 0)
    for k in sorted(acc):
        big = {d: t for d, t in dur[k].items() if t > 500_000}  # the full passes (skipped launches return at once)
        if not big:
            continue
        ds = sorted(big, key=lambda d: big[d])
        med = ds[len(ds) // 2]
        out.append(f"[{sub}] {k}: {len(big)} full passes, median duration {big[med]} ns")
        for c in sorted(acc[k]):
            v = acc[k][c].get(med, 0.0)
            extra = ""
            if c == "FETCH_SIZE":
                extra = f"  -> {v * 1024 * 2 / 1e9:.3f} GB corrected (x2, gfx950) vs 9.600 GB algorithmic"
            out.append(f"    {c} = {int(v)}{extra}")
        c = {n: acc[k][n].get(med, 0.0) for n in acc[k]}
        if "SQ_LDS_IDX_ACTIVE" in c and "SQ_LDS_BANK_CONFLICT" in c and big[med] > 0:
            # LDS array cycles per CU = IDX_ACTIVE / 256; the kernel's duration in shader cycles at ~2.4 GHz
            per_cu = c["SQ_LDS_IDX_ACTIVE"] / 256.0
            out.append(f"    derived: LDS-array cycles per CU = {per_cu:.3e} = {per_cu / 2.4e9 * 1e3:.3f} ms at 2.4 GHz of a {big[med] / 1e6:.3f} ms pass "
                       f"({per_cu / 2.4e9 * 1e9 / big[med] * 100:.0f} % busy); conflict share of LDS cycles = {c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.2f}")
        if "SQ_INSTS_LDS" in c and "SQ_INSTS_VALU" in c:
            out.append(f"    derived: per 64-row tile: {c['SQ_INSTS_LDS'] / (1e8 / 64):.0f} LDS instructions, {c['SQ_INSTS_VALU'] / (1e8 / 64):.0f} VALU instructions")
if len(out) > 3:
    open(os.path.join(DST, f"{R}_pmc_pq.txt"), "w").write("\n".join(out) + "\n")
    print("pq counters ->", f"{R}_pmc_pq.txt")
