#!/usr/bin/env python3
"""Export wall-clock as SURVEY.md 8(d) defines it -- "dcsg_export from scene files on disk to both output files closed"
-- measured in a FRESH process so that nothing is warm that a host starting an export would not have:

    python tools/export_from_disk.py --scene-dir tests/golden/scenes/design2 --levels 7 9 10 [--cold-jit]

Prints one JSON line: seconds of (1) importing the binding and loading libdcsg_hip, (2) dcsg_create (HIP runtime +
device + built-in code object), (3) reading the scene files + dcsg_load_scene (parse, code object from the JIT cache, or
compiled when --cold-jit sets DCSG_JIT_NO_CACHE=1 and AMD_COMGR_CACHE=0, module load, uploads), (4) the first dcsg_export (creates the output
files, first-touch of every buffer) and (5) the best of the following exports (steady state: files overwritten in
place); plus the library's own stage breakdown of (2) and (3).  `disk_to_files_s` = (3) + (4): what pressing "export" in
a running host costs the first time; `with_context_s` adds (2).
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
    ap.add_argument("--scene-dir", required=True)
    ap.add_argument("--levels", type=int, nargs=3, default=None)
    ap.add_argument("--cold-jit", action="store_true")
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    if args.cold_jit:
        os.environ["DCSG_JIT_NO_CACHE"] = "1"
        os.environ["AMD_COMGR_CACHE"] = "0"   # libamd_comgr keeps a cache of its own (~/.cache/comgr): cold means cold
    t0 = time.perf_counter()
    from designcsg_amd.evaluator import Evaluator
    from designcsg_amd import _lib
    _lib.lib()
    t_import = time.perf_counter() - t0
    t0 = time.perf_counter()
    ev = Evaluator(0)
    t_create = time.perf_counter() - t0
    t0 = time.perf_counter()
    status, log = ev.build_from_dir(args.scene_dir)
    lines = open(os.path.join(args.scene_dir, "exportConfig.txt")).read().split("\n")
    t_load = time.perf_counter() - t0
    if status != 0:
        sys.exit(log)
    if args.levels:
        lines[1:4] = [str(v) for v in args.levels]
    cfg = "\n".join(lines)
    walls, stages = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for _ in range(1 + max(1, args.repeat)):
            t0 = time.perf_counter()
            st = ev.export(cfg, os.path.join(tmp, "o.stl"), os.path.join(tmp, "o.ply"))
            walls.append(time.perf_counter() - t0)
            stages.append({k: round(getattr(st, "t_" + k) * 1e3, 3) for k in ("bbox", "sweep", "mesh", "project", "download", "write_stl", "write_ply", "total")})
        sizes = [os.path.getsize(os.path.join(tmp, f)) for f in ("o.stl", "o.ply")]
    prof = ev.load_profile()
    out = {"scene": os.path.basename(os.path.normpath(args.scene_dir)), "levels": [int(v) for v in lines[1:4]],
           "jit_cache": "cold (DCSG_JIT_NO_CACHE=1: compiled)" if args.cold_jit else "warm",
           "triangles": int(st.triangles), "stl_bytes": sizes[0], "ply_bytes": sizes[1],
           "import_and_load_library_s": round(t_import, 4), "create_context_s": round(t_create, 4),
           "read_files_and_load_scene_s": round(t_load, 4), "first_export_s": round(walls[0], 4),
           "steady_export_s": round(min(walls[1:]), 4),
           "disk_to_files_s": round(t_load + walls[0], 4), "with_context_s": round(t_create + t_load + walls[0], 4),
           "first_export_stages_ms": stages[0], "steady_export_stages_ms": stages[walls.index(min(walls[1:]))],
           "create_ms": prof["create"], "load_scene_ms": prof["load_scene"]}
    print(json.dumps(out))
    ev.close()


if __name__ == "__main__":
    main()
