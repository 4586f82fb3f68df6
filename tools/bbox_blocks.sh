#!/bin/bash
# Stage A against the number of blocks per CU (every block commits its result with atomics on the same few words)
for n in 2 4 8 16 32; do
  for s in design2 sphere flat256; do
    DCSG_BBOX_BLOCKS_PER_CU=$n python tools/bench_export.py --scene $s --levels 6 8 9 --repeat 5 > gpurun_out/bbb.json || exit 1
    python - $n $s <<'PY'
import json, sys
d = json.load(open("gpurun_out/bbb.json"))
print("blocks/CU %-3s %-9s stage A ms: %s" % (sys.argv[1], sys.argv[2], " ".join("%.3f" % (r["bbox"] * 1e3) for r in d["gpu_runs"])))
PY
  done
done
