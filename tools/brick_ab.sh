#!/bin/bash
# Dense culling sweeps (fused and SDF only) of the 256-brush scenes at 513^3 against the brick shape (DCSG_BRICK=wide,depth)
for b in "0,4" "0,8" "1,2" "1,4" "1,8" "1,16"; do
  for w in flat256_512 grouped256_512 smooth256_512; do
    DCSG_BRICK=$b python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-export > gpurun_out/brick_ab.json 2>/dev/null || exit 1
    python - $b $w <<'PY'
import json, sys
d = json.load(open("gpurun_out/brick_ab.json"))
print("brick %-5s %-15s fused %.3f ms   SDF only %.3f ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["sdf_only_sweep"]["kernel_ms"]))
PY
  done
done
