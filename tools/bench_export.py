#!/usr/bin/env python3
"""Export wall-clock (BASELINE metric "export wall-clock s"): scene files on disk -> .stl and .ply closed.

    python tools/bench_export.py --scene design2 --levels 7 9 10 [--cpu] [--dense-normals] [--repeat 3]

Prints one JSON line with the per-stage seconds of dcsg_export (stage A, lattice sweep, mesher, projection,
download, writers).  --cpu also times the oracle's pipeline (own CPU restatement: lazily sampled lattice,
OpenMP) on this host for the same configuration -- a reported baseline, see bench.py for the headline metric.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="design2")
    ap.add_argument("--levels", type=int, nargs=3, default=None, metavar=("MIN", "MAX", "GRID"))
    ap.add_argument("--gd", type=int, default=None)
    ap.add_argument("--dense-normals", action="store_true")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()

    from designcsg_amd.evaluator import Evaluator
    d = os.path.join(ROOT, "tests", "golden", "scenes", args.scene)
    lines = open(os.path.join(d, "exportConfig.txt")).read().split("\n")
    if args.levels:
        lines[1:4] = [str(v) for v in args.levels]
    if args.gd is not None:
        lines[5] = str(args.gd)
    cfg_txt = "\n".join(lines)
    out = {"scene": args.scene, "levels": [int(v) for v in lines[1:4]], "gd_steps": int(lines[5]),
           "dense_normals": args.dense_normals}
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        ev = Evaluator(0)
        status, log = ev.build_from_dir(d)
        assert status == 0, log
        out["t_create_and_load_scene"] = round(time.perf_counter() - t0, 4)
        out["scene_info"] = ev.scene_info()
        runs = []
        for _ in range(args.repeat):
            t0 = time.perf_counter()
            st = ev.export(cfg_txt, os.path.join(tmp, "o.stl"), os.path.join(tmp, "o.ply"), dense_normals=args.dense_normals)
            wall = time.perf_counter() - t0
            runs.append({"wall": round(wall, 4), "bbox": round(st.t_bbox, 4), "sweep": round(st.t_sweep, 4),
                         "mesh": round(st.t_mesh, 4), "project": round(st.t_project, 4),
                         "download": round(st.t_download, 4), "write_stl": round(st.t_write_stl, 4),
                         "write_ply": round(st.t_write_ply, 4), "sweep_kernel_ms": round(st.sweep_kernel_ms, 3)})
            runs[-1]["cells"] = ev.debug_cells()
        out["triangles"] = int(st.triangles)
        out["lattice_points"] = int(st.lattice_points)
        out["stl_bytes"] = os.path.getsize(os.path.join(tmp, "o.stl"))
        out["ply_bytes"] = os.path.getsize(os.path.join(tmp, "o.ply"))
        out["gpu_runs"] = runs
        out["gpu_export_s"] = min(r["wall"] for r in runs)
        ev.close()
        if args.cpu:
            from oracle import oracle as O
            O.build()
            sc = O.Scene.from_dir(d)
            cfg = O.parse_export_config(cfg_txt)
            tab = O.LookupTable(open(os.path.join(ROOT, "designcsg_amd", "data", "lookupTable.txt")).read())
            th = args.cpu_threads
            t0 = time.perf_counter()
            c, dia, _ = sc.estimate_bounding_box(cfg.bounding_box_half_diameter, threads=th)
            t1 = time.perf_counter()
            tris, _ = sc.get_surface(c, dia[0], cfg, tab, threads=th)
            t2 = time.perf_counter()
            tris = sc.gradient_descent(tris, cfg.gradient_descent_steps, threads=th)
            t3 = time.perf_counter()
            O.write_stl(os.path.join(tmp, "c.stl"), tris)
            O.write_ply(os.path.join(tmp, "c.ply"), tris)
            t4 = time.perf_counter()
            out["cpu_oracle"] = {"threads": th, "wall": round(t4 - t0, 3), "bbox": round(t1 - t0, 3),
                                 "mesh": round(t2 - t1, 3), "project": round(t3 - t2, 3), "write": round(t4 - t3, 3),
                                 "triangles": int(len(tris))}
            out["speedup_vs_cpu_oracle"] = round(out["cpu_oracle"]["wall"] / out["gpu_export_s"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
