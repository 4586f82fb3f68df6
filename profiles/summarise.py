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
    for k, o in out.items():
        c = o["counters_mean_per_launch"]
        if "WRITE_SIZE" in c or "FETCH_SIZE" in c:
            w = c.get("WRITE_SIZE", 0.0) * 1024.0
            r = c.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0  # gfx950: FETCH_SIZE counts 64 B per 128-B request
            o["hbm_bytes_per_launch"] = {"written": w, "read_corrected": r, "total": w + r}
    json.dump(out, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
