#!/usr/bin/env python3
"""Turn the rocprofv3 output of scripts/gpu_profile.sh (gpurun_out/prof_<tag>/) into the committed evidence under
profiles/: per-kernel stats (kernel-trace --stats), per-launch HBM traffic from the PMC passes, traffic.json for
bench.py.  usage: python scripts/summarize_profiles.py <tag> [pattern]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def newest(pattern):
    """gpurun_out/ keeps the files of earlier calls: only the most recent run of a directory counts"""
    fs = glob.glob(pattern, recursive=True)
    return [max(fs, key=os.path.getmtime)] if fs else []


def pmc(path):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest(os.path.join(path, "**", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    pattern = sys.argv[2] if len(sys.argv) > 2 else "banded"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    lines = ["# rocprofv3 summary %s (%s)" % (tag, pattern), ""]
    # ---- kernel stats of the default bench command -------------------------------------------------------
    stats = newest(os.path.join(src, "stats", "**", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, pattern)))
        rows = list(csv.DictReader(open(stats[0])))
        lines += ["## `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu` (whole run: SpMM leg + %s)" % "solve", "",
                  "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
        for r in rows[:16]:
            lines.append("| %s | %s | %.2f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                           float(r["AverageNs"]) / 1e3, r["Percentage"]))
        lines.append("")
    for f in ("bench_stats.json",):
        p = os.path.join(src, f)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, "%s_%s_bench_under_rocprof.json" % (tag, pattern)))
    # ---- PMC: HBM traffic per launch ----------------------------------------------------------------------
    traffic = {}
    fetch, write = pmc(os.path.join(src, "pmc_FETCH_SIZE")), pmc(os.path.join(src, "pmc_WRITE_SIZE"))
    fs, ws = pmc(os.path.join(src, "pmcsolve_FETCH_SIZE")), pmc(os.path.join(src, "pmcsolve_WRITE_SIZE"))
    lines += ["## HBM traffic per launch from PMC passes (`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate runs)", "",
              "FETCH_SIZE / WRITE_SIZE are in KiB.  WRITE_SIZE is exact for streaming stores (calibrated here on `k_random`: 1 000 000 KiB",
              "for a 1M x 128 fp64 fill).  On gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced streaming reads",
              "(MI355X_MICROARCH.md, HBM section): the `fetch x2` column applies that correction; it is calibrated below on the Lanczos",
              "pass, whose read volume is known exactly.", "",
              "| kernel | launches | FETCH_SIZE KiB | fetch x2 GB | WRITE_SIZE KiB | write GB | traffic GB (2*fetch + write) |", "|---|---|---|---|---|---|---|"]
    names = sorted(set(fetch) | set(write) | set(fs) | set(ws))
    best = {}
    for k in names:
        fv = fetch.get(k, {}).get("FETCH_SIZE") or fs.get(k, {}).get("FETCH_SIZE") or []
        wv = write.get(k, {}).get("WRITE_SIZE") or ws.get(k, {}).get("WRITE_SIZE") or []
        if not fv and not wv:
            continue
        fa = sum(fv) / len(fv) if fv else 0.0
        wa = sum(wv) / len(wv) if wv else 0.0
        tot = (2 * fa + wa) * 1024 / 1e9
        lines.append("| %s | %d | %.0f | %.3f | %.0f | %.3f | %.3f |" % (k, max(len(fv), len(wv)), fa, 2 * fa * 1024 / 1e9, wa, wa * 1024 / 1e9, tot))
        # roofline.traffic: the A*X kernel of the spmm-only PMC pass (128 columns), i.e. the one bench.py times
        if "spmm" in k and (k in fetch or k in write):
            n_here = max(len(fetch.get(k, {}).get("FETCH_SIZE", [])), len(write.get(k, {}).get("WRITE_SIZE", [])))
            key = "%s:%s" % (k.split("<")[0], pattern)
            if n_here >= best.get(key, (0, 0))[0]:
                fa2 = fetch.get(k, {}).get("FETCH_SIZE", [0.0])
                wa2 = write.get(k, {}).get("WRITE_SIZE", [0.0])
                best[key] = (n_here, (2 * sum(fa2) / len(fa2) + sum(wa2) / len(wa2)) * 1024, k)
    lines.append("")
    traffic = {k: v[1] for k, v in best.items()}
    # the kernel's average duration in the stats pass of the same build: bench.py compares it with what it measures live and says so in
    # `roofline.traffic_stale` when the kernel has changed since these passes were made
    if stats:
        for key in list(traffic):
            kname = key.split(":")[0]
            # the instantiation the PMC pass saw (template arguments and all: a kernel family has other widths in the same run); the
            # family's most-called member only when the stats pass does not have that one
            durs = [(int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in rows if short(r["Name"]) == best[key][2]]
            if not durs:
                durs = [(int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in rows if short(r["Name"]).split("<")[0] == kname]
            if durs:
                traffic[key + ":kernel_us"] = max(durs)[1]
    json.dump(traffic, open(os.path.join(dst, "traffic_%s_%s.json" % (tag, pattern)), "w"), indent=1)
    # merge into profiles/traffic.json (what bench.py reads)
    tj = os.path.join(dst, "traffic.json")
    cur = json.load(open(tj)) if os.path.exists(tj) else {}
    cur.update(traffic)
    json.dump(cur, open(tj, "w"), indent=1, sort_keys=True)
    open(os.path.join(dst, "%s_%s_summary.md" % (tag, pattern)), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
