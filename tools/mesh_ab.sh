#!/bin/bash
# Stage times of four 1024^3 exports, as tools/c5_exports.sh prints them; extra environment in front selects the variant:
#   DCSG_DENSE_NORMALS=1 bash tools/mesh_ab.sh dense
TAG=${1:-mesh}
for s in design2 flat256 grouped256 smooth256; do
  python tools/bench_export.py --scene $s --levels 7 9 10 --repeat 5 > gpurun_out/${TAG}_${s}_1024.json || exit 1
done
python - "$TAG" <<'PY'
import glob, json, sys
for f in sorted(glob.glob("gpurun_out/%s_*_1024.json" % sys.argv[1])):
    d = json.load(open(f))
    best = min(d["gpu_runs"], key=lambda r: r["wall"])
    print("%-28s tris %8d wall %.4f (first %.4f) bbox %.4f sweep %.4f mesh %.4f project %.4f" % (
        f.split("/")[-1], d["triangles"], best["wall"], d["gpu_runs"][0]["wall"], best["bbox"], best["sweep"], best["mesh"], best["project"]))
PY
