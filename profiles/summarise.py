#!/usr/bin/env python3
"""Reduce gpurun_out/prof_<tag>/ (written by profiles/run_profile.sh on the GPU box) to the small files
that are committed under profiles/:  <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats),
<tag>_pmc_summary.json (per-kernel means of every counter, per launch), <tag>_bench.json (the bench line of
the profiled run).  HBM traffic follows MI355X_MICROARCH.md: WRITE_SIZE / FETCH_SIZE are in KiB-like units of
1024 B per count... (rocprofv3 reports KB); FETCH_SIZE is doubled on gfx950 for wide coalesced reads."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    shutil.copyfile(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
    shutil.copyfile(os.path.join(src, "stats_bench.json"), os.path.join(dst, tag + "_bench.json"))
    # rocprofv3's --stats averages mix the launches from idle clocks (scene loading leaves the GPU idle; bench.py's
    # from_idle figure) with the settled ones: per kernel also the median, the minimum and the mean of the LAST `steps`
    # launches -- the timed region of the bench line, which is what `ms_per_step` and `roofline.kernel_ms` are taken over
    trace = os.path.join(src, "stats", "stats_kernel_trace.csv")
    if os.path.exists(trace):
        try:
            steps = int(json.load(open(os.path.join(src, "stats_bench.json"))).get("steps", 20))
        except (OSError, ValueError):
            steps = 20
        dur = collections.defaultdict(list)
        rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        med = {}
        for k, v in dur.items():
            if not k.startswith("dcsg_"):
                continue
            sv = sorted(v)
            med[k] = {"launches": len(v), "mean_ms": round(sum(v) / len(v), 5), "median_ms": round(sv[len(sv) // 2], 5), "min_ms": round(sv[0], 5),
                      "max_ms": round(sv[-1], 5), "timed_region_mean_ms": round(sum(v[-steps:]) / len(v[-steps:]), 5), "timed_region_launches": len(v[-steps:])}
        json.dump(med, open(os.path.join(dst, tag + "_kernel_medians.json"), "w"), indent=1, sort_keys=True)
    out = {}
    for sub, name in (("pmc_sq", "sq"), ("pmc_sq2", "sq2"), ("pmc_write", "wr"), ("pmc_fetch", "rd")):
        f = os.path.join(src, sub, name + "_counter_collection.csv")
        if not os.path.exists(f):
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not k.startswith("dcsg_"):
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]), "vgpr": int(r["VGPR_Count"]),
                       "sgpr": int(r["SGPR_Count"]), "scratch": int(r["Scratch_Size"]), "lds": int(r["LDS_Block_Size"])}
        for k, counters in acc.items():
            o = out.setdefault(k, {"launch": meta[k], "counters_mean_per_launch": {}, "dispatches": 0})
            for c, v in counters.items():
                o["counters_mean_per_launch"][c] = sum(v) / len(v)
                o["dispatches"] = max(o["dispatches"], len(v))
    # A sweep of more than 0xf0000000 lattice points is several launches of the same kernel (bench.py:
    # config.launches_per_sweep; 1 for every workload up to 1025^3 since round 3): per-SWEEP figures are the per-launch
    # means times that count, so that they can stand next to the algorithmic bytes of a whole sweep.
    try:
        lps = int(json.load(open(os.path.join(src, "stats_bench.json"))).get("config", {}).get("launches_per_sweep", 1))
    except (OSError, ValueError):
        lps = 1
    for k, o in out.items():
        c = o["counters_mean_per_launch"]
        o["launches_per_sweep"] = lps if k.startswith("dcsg_sweep") else 1
        if "WRITE_SIZE" in c or "FETCH_SIZE" in c:
            w = c.get("WRITE_SIZE", 0.0) * 1024.0
            r = c.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0  # gfx950: FETCH_SIZE counts 64 B per 128-B request
            o["hbm_bytes_per_launch"] = {"written": w, "read_corrected": r, "total": w + r}
            n = o["launches_per_sweep"]
            o["hbm_bytes_per_sweep"] = {"written": w * n, "read_corrected": r * n, "total": (w + r) * n}
    json.dump(out, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
