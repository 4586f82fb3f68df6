#!/bin/bash
# Projection of per-scene code: a lane per stencil point (default) against a vertex per lane (DCSG_PROJECT_VERTEX_PER_LANE=1)
for v in "" "DCSG_PROJECT_VERTEX_PER_LANE=1"; do
  for cfg in "sphere 6 8 9" "design2 6 8 9" "design2 7 9 10" "design1 6 8 9" "userbrush128 6 8 9"; do
    set -- $cfg
    env $v python tools/bench_export.py --scene $1 --levels $2 $3 $4 --repeat 5 > gpurun_out/pab.json || exit 1
    python - "$v" "$cfg" <<'PY'
import json, sys
d = json.load(open("gpurun_out/pab.json"))
r = sorted(d["gpu_runs"][1:], key=lambda r: r["wall"])
print("%-32s %-20s wall best %.4f median %.4f | project stage ms: %s" % (sys.argv[1] or "lane per stencil point", sys.argv[2], r[0]["wall"], r[len(r) // 2]["wall"], " ".join("%.2f" % (x["project"] * 1e3) for x in d["gpu_runs"])))
PY
  done
done
