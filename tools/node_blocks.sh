#!/bin/bash
# Node-wise sweep against the number of blocks per CU that walk the node list (0 = a block per node / group of eight)
for n in 0 16 32 64 128; do
  for cfg in "design2 7 9 10" "design2 6 8 9" "flat256 7 9 10" "grouped256 7 9 10"; do
    set -- $cfg
    DCSG_NODE_BLOCKS_PER_CU=$n python tools/bench_export.py --scene $1 --levels $2 $3 $4 --repeat 4 > gpurun_out/nb.json || exit 1
    python - $n "$cfg" <<'PY'
import json, sys
d = json.load(open("gpurun_out/nb.json"))
print("blocks/CU %-4s %-20s sweep kernel ms: %s" % (sys.argv[1], sys.argv[2], " ".join("%.3f" % r["sweep_kernel_ms"] for r in d["gpu_runs"])))
PY
  done
done
