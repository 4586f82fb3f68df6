#!/bin/bash
# Stage times of the Design2 / sphere / userbrush exports: one line per scene and size.  Extra environment in front.
TAG=${1:-d2}
for cfg in "design2 6 8 9" "design2 7 9 10" "sphere 6 8 9" "design1 6 8 9" "userbrush128 6 8 9"; do
  set -- $cfg
  python tools/bench_export.py --scene $1 --levels $2 $3 $4 --repeat 5 > gpurun_out/${TAG}_$1_$((1 << $4)).json || exit 1
done
python - "$TAG" <<'PY'
import glob, json, sys
for f in sorted(glob.glob("gpurun_out/%s_*.json" % sys.argv[1])):
    d = json.load(open(f))
    best = min(d["gpu_runs"], key=lambda r: r["wall"])
    print("%-28s tris %8d wall %.4f bbox %.4f sweep %.4f mesh %.4f project %.4f download %.4f stl %.4f ply %.4f" % (
        f.split("/")[-1], d["triangles"], best["wall"], best["bbox"], best["sweep"], best["mesh"], best["project"], best["download"], best["write_stl"], best["write_ply"]))
PY
