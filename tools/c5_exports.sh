#!/bin/bash
# Stage times of the many-object exports (BASELINE config 5) at 512^3 and 1024^3: one JSON line per scene and size.
# Extra environment (A/B of one switch) goes in front:  DCSG_PROJECT_WAVE_LISTS=1 bash tools/c5_exports.sh tag
TAG=${1:-c5}
for s in flat256 grouped256 smooth256; do
  for l in "6 8 9" "7 9 10"; do
    set -- $l
    python tools/bench_export.py --scene $s --levels $l --repeat 5 > gpurun_out/${TAG}_${s}_$((1 << $3)).json || exit 1
  done
done
python - "$TAG" <<'PY'
import glob, json, sys
for f in sorted(glob.glob("gpurun_out/%s_*.json" % sys.argv[1])):
    d = json.load(open(f))
    best = min(d["gpu_runs"], key=lambda r: r["wall"])
    print("%-28s tris %8d wall %.4f bbox %.4f sweep %.4f mesh %.4f project %.4f (projection over the runs: %s) cells %s then %s" % (
        f.split("/")[-1], d["triangles"], best["wall"], best["bbox"], best["sweep"], best["mesh"], best["project"],
        " ".join("%.4f" % r["project"] for r in d["gpu_runs"]), d["gpu_runs"][0].get("cells"), d["gpu_runs"][-1].get("cells")))
PY
