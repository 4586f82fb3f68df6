#!/bin/bash
# Steady export wall time and writer times against the number of writer threads per file
for s in design2 flat256; do
  for n in 2 4 6 8 12; do
    DCSG_WRITER_THREADS=$n python tools/bench_export.py --scene $s --levels 7 9 10 --repeat 6 > gpurun_out/wt.json || exit 1
    python - $s $n <<'PY'
import json, sys
d = json.load(open("gpurun_out/wt.json"))
r = sorted(d["gpu_runs"][1:], key=lambda r: r["wall"])
m = r[len(r) // 2]
print("%-8s threads %2s  wall median %.4f best %.4f  stl %.4f ply %.4f" % (sys.argv[1], sys.argv[2], m["wall"], r[0]["wall"], m["write_stl"], m["write_ply"]))
PY
  done
done
