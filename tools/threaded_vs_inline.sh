#!/bin/bash
# Per-scene code of a scene of its own brushes: inline (default up to 48 evaluations) against the threaded form
# (DCSG_SPECIALISE=1 makes every such scene threaded): cold load, dense sweeps, export.
for s in design2 design1 const_table; do
  for mode in inline threaded; do
    if [ $mode = threaded ]; then export DCSG_SPECIALISE=1; else unset DCSG_SPECIALISE; fi
    echo "== $s $mode"
    python tools/export_from_disk.py --scene-dir tests/golden/scenes/$s --levels 6 8 9 --cold-jit | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cold: load', d['read_files_and_load_scene_s'], 'first export', d['first_export_s'], 'steady', d['steady_export_s'], d['load_scene_ms'])"
    python - $s <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import bench
from designcsg_amd.evaluator import Evaluator
name = sys.argv[1]
ev = Evaluator(0)
assert ev.build_from_dir(os.path.join("tests", "golden", "scenes", name))[0] == 0
c, d, _ = ev.estimate_bounding_box(10.0)
t = bench.time_sweeps(ev, c, d, 512, reps=3, settle_s=0.1)
print("sweeps 513^3: fused %.3f ms, sdf-only %.3f ms" % (t["fused"]["kernel_ms"], t["sdf_only"]["kernel_ms"]))
ev.close()
PY
  done
done
unset DCSG_SPECIALISE
