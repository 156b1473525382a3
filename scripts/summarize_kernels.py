#!/usr/bin/env python3
"""gpurun_out/kern (scripts/gpu_kernels.sh) -> profiles/<tag>_kernels.md: per-kernel microbenchmark, rocprofv3 kernel stats of the same
command, MFMA counters of the projection kernels.  usage: python scripts/summarize_kernels.py [tag]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "kern")
out = ["# Per-kernel evidence at the C3 sizes (m = 1M rows, k = 200, p = w = 16) -- `scripts/gpu_kernels.sh`", "",
       "## HIP-event time per call through the C ABI (`python3 scripts/kernel_bench.py`, median of 10)", "",
       "`GB/s` and `frac` are ALGORITHMIC bytes (SURVEY.md 8(d) formulas) / time against 8 TB/s; the first row is what a plain streaming",
       "copy reaches on the same box (the practical ceiling for the bandwidth-bound kernels).  Calls that return a host value include the",
       "D2H copy and the stream synchronisation.", "",
       "| case | ms | alg. GB | GB/s | frac of 8 TB/s | TFLOP/s |", "|---|---|---|---|---|---|"]
for l in open(os.path.join(src, "kernel_bench.jsonl")):
    d = json.loads(l)
    if "case" in d:
        out.append("| %s | %.3f | %.3f | %.0f | %.3f | %s |" % (d["case"], d["ms"], d["alg_GB"], d["GBs"], d["frac_hbm_8TBs"], ("%.1f" % d["TFLOPs"]) if d["GFLOP"] else "--"))
out += ["", "## `rocprofv3 --kernel-trace --stats -- python3 scripts/kernel_bench.py --reps 5`", "", "| kernel | calls | avg us | total ms |", "|---|---|---|---|"]


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def newest(pattern):
    fs = glob.glob(pattern, recursive=True)
    return [max(fs, key=os.path.getmtime)] if fs else []


for f in newest(os.path.join(src, "stats", "**", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(ROOT, "profiles", "%s_kernels_kernel_stats.csv" % tag))
    for r in list(csv.DictReader(open(f)))[:20]:
        out.append("| %s | %s | %.1f | %.2f |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
acc = collections.defaultdict(lambda: collections.defaultdict(dict))
for f in newest(os.path.join(src, "pmc_mfma", "**", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
out += ["", "## MFMA utilisation of the projection kernels (`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE ...`)", "",
        "`SQ_VALU_MFMA_BUSY_CYCLES` counts cycles per SIMD, `GRBM_GUI_ACTIVE` is summed over the 8 XCDs: utilisation = MFMA busy cycles /",
        "(GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs).  One `v_mfma_f64_16x16x4` occupies its SIMD for 16 cycles.  Grouped by launch shape",
        "(distinct MFMA op counts = distinct problem sizes of the microbenchmark).", "",
        "| kernel | MFMA ops F64 (x512 flop) | launches | GUI cycles/XCD | MFMA busy | utilisation |", "|---|---|---|---|---|---|"]
for k in sorted(acc):
    if not ("gram" in k or "gemm" in k):
        continue
    groups = collections.defaultdict(list)
    for disp, c in acc[k].items():
        groups[c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0)].append(c)
    for mops in sorted(groups):
        cs = groups[mops]
        gui = sum(c.get("GRBM_GUI_ACTIVE", 0) for c in cs) / len(cs) / 8
        busy = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for c in cs) / len(cs)
        out.append("| %s | %.3g | %d | %.3g | %.3g | %.1f %% |" % (k, mops, len(cs), gui, busy, 100 * busy / (gui * 1024) if gui else 0))
out.append("")
open(os.path.join(ROOT, "profiles", "%s_kernels.md" % tag), "w").write("\n".join(out) + "\n")
print("\n".join(out))
