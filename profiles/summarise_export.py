#!/usr/bin/env python3
"""Reduce gpurun_out/prof_<tag>/ (profiles/export_profile.sh) to profiles/<tag>.json and profiles/<tag>.txt: per kernel of
an export -- launches per export, duration (mean over all exports of the traced run, and the SETTLED median: exports 2..5,
the first export of a process being the cold one), share of the export's GPU time, waves, VALU / SALU / LDS instructions per
wave, VALU instructions per SIMD and core cycle (SQ_INSTS_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); the issue
ceiling is 0.5: one wave64 VALU instruction per SIMD every other cycle), the same as a fraction of bench.py's fixed VALU
ceiling (1.229e12 wave-instructions/s), and HBM bytes per launch (WRITE_SIZE / FETCH_SIZE x 1024, FETCH_SIZE doubled on
gfx950: MI355X_MICROARCH.md) against the 8 TB/s peak."""
import collections
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VALU_CEILING = 1.229e12   # wave64 VALU instructions per second, whole chip (bench.py)
HBM_PEAK = 8.0e12


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    f = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return acc
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    rows = list(csv.DictReader(open(glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    export = json.load(open(os.path.join(src, "export.json")))
    n_exports = len(export["gpu_runs"])
    # exports are delimited by the stage-A kernel (one launch per export)
    bbox = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("dcsg_bbox")]
    assert len(bbox) == n_exports, (len(bbox), n_exports)
    per_export = []
    for e in range(n_exports):
        chunk = rows[bbox[e]:bbox[e + 1] if e + 1 < n_exports else len(rows)]
        d = collections.defaultdict(list)
        for r in chunk:
            d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        per_export.append(d)
    sq, wr, rd = counters(os.path.join(src, "pmc_sq")), counters(os.path.join(src, "pmc_write")), counters(os.path.join(src, "pmc_fetch"))
    kernels = sorted({k for d in per_export for k in d if k.startswith("dcsg_")})
    settled_total = statistics.median([sum(sum(v) for k, v in d.items() if k.startswith("dcsg_")) for d in per_export[1:]])
    out = {"scene": export["scene"], "levels": export["levels"], "triangles": export["triangles"], "exports_traced": n_exports,
           "gpu_ms_per_export_settled": round(settled_total, 4), "kernels": {}}
    for k in kernels:
        per = [sum(d.get(k, [])) for d in per_export]                  # ms per export, all launches of the kernel
        launches = statistics.median([len(d.get(k, [])) for d in per_export])
        o = {"launches_per_export": launches, "ms_per_export_first": round(per[0], 4), "ms_per_export_mean": round(sum(per) / len(per), 4),
             "ms_per_export_settled_median": round(statistics.median(per[1:]), 4)}
        o["share_of_export_gpu_time"] = round(o["ms_per_export_settled_median"] / settled_total, 4) if settled_total else None
        m = {c: v for c, v in sq.get(k, {}).items()}
        if m.get("SQ_WAVES"):
            # counters per launch: use the launches of the settled exports (drop the first export's share)
            def settled(c):
                v = m[c]
                per_exp = max(1, int(len(v) / 3))
                v = v[per_exp:] if len(v) > per_exp else v
                return sum(v) / len(v)
            waves, valu, salu, lds = settled("SQ_WAVES"), settled("SQ_INSTS_VALU"), settled("SQ_INSTS_SALU"), settled("SQ_INSTS_LDS")
            cyc = settled("GRBM_GUI_ACTIVE") / 8.0
            ms_launch = o["ms_per_export_settled_median"] / max(1, launches)
            o.update({"waves_per_launch": round(waves), "valu_per_wave": round(valu / waves), "salu_per_wave": round(salu / waves),
                      "lds_per_wave": round(lds / waves), "core_cycles_per_launch": round(cyc),
                      "valu_per_SIMD_cycle": round(valu / 1024.0 / cyc, 4) if cyc else None,
                      "wave_cycles_per_wave": round(4 * settled("SQ_WAVE_CYCLES") / waves),
                      "valu_frac_of_ceiling": round(valu / (ms_launch * 1e-3) / VALU_CEILING, 4) if ms_launch else None})
        w = wr.get(k, {}).get("WRITE_SIZE")
        r = rd.get(k, {}).get("FETCH_SIZE")
        if w or r:
            wb = sum(w) / len(w) * 1024.0 if w else 0.0
            rb = sum(r) / len(r) * 1024.0 * 2.0 if r else 0.0
            ms_launch = o["ms_per_export_settled_median"] / max(1, launches)
            o["hbm_bytes_per_launch"] = {"written": round(wb), "read_corrected": round(rb)}
            o["hbm_frac_of_peak"] = round((wb + rb) / (ms_launch * 1e-3) / HBM_PEAK, 4) if ms_launch else None
        out["kernels"][k] = o
    dom = max(out["kernels"], key=lambda k: out["kernels"][k]["ms_per_export_settled_median"])
    out["dominant_kernel"] = dom
    json.dump(out, open(os.path.join(ROOT, "profiles", tag + ".json"), "w"), indent=1, sort_keys=True)
    with open(os.path.join(ROOT, "profiles", tag + ".txt"), "w") as f:
        f.write("# %s levels %s: %d triangles; GPU ms per export (settled median of exports 2..%d): %.3f\n" % (
            out["scene"], out["levels"], out["triangles"], n_exports, settled_total))
        f.write("%-26s %5s %9s %9s %9s %6s %9s %9s %9s %8s %8s %8s\n" % ("kernel", "n", "first_ms", "mean_ms", "settled", "share", "waves", "VALU/wave",
                                                                        "SALU/wave", "VALU/cyc", "VALUfrac", "HBMfrac"))
        for k, o in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["ms_per_export_settled_median"]):
            f.write("%-26s %5s %9.4f %9.4f %9.4f %6.3f %9s %9s %9s %8s %8s %8s\n" % (
                k[:26], o["launches_per_export"], o["ms_per_export_first"], o["ms_per_export_mean"], o["ms_per_export_settled_median"],
                o["share_of_export_gpu_time"] or 0, o.get("waves_per_launch", "-"), o.get("valu_per_wave", "-"), o.get("salu_per_wave", "-"),
                o.get("valu_per_SIMD_cycle", "-"), o.get("valu_frac_of_ceiling", "-"), o.get("hbm_frac_of_peak", "-")))
    print(open(os.path.join(ROOT, "profiles", tag + ".txt")).read())


if __name__ == "__main__":
    for t in sys.argv[1:]:
        main(t)
