"""Diagnostic: spatial extent of the runs of 64 consecutive unique vertices the projection kernel works on
(python tools/cluster_probe.py [scene] [min max grid])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from designcsg_amd.evaluator import Evaluator

scene = sys.argv[1] if len(sys.argv) > 1 else "flat256"
levels = [int(v) for v in sys.argv[2:5]] if len(sys.argv) > 4 else [6, 8, 9]
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes", scene)
lines = open(os.path.join(d, "exportConfig.txt")).read().split("\n")
lines[1:4] = [str(v) for v in levels]
lines[5] = "0"
ev = Evaluator(0)
assert ev.build_from_dir(d)[0] == 0
st = ev.export("\n".join(lines))
t = ev.download_triangles().reshape(-1, 3)
cell = st.box_diameter / (1 << levels[2])
key = np.ascontiguousarray(t).view([("", np.uint32)] * 3).ravel()
_, first = np.unique(key, return_index=True)
u = t[np.sort(first)]                      # unique vertices in order of first appearance
n = len(u) // 64 * 64
c = u[:n].reshape(-1, 64, 3)
ext = (c.max(1) - c.min(1)) / cell         # bounding-box extent of each run, in cells
hd = 0.5 * np.linalg.norm(ext, axis=1)
print(scene, levels, "triangles", len(t) // 3, "unique vertices", len(u))
print("run half-diagonal in cells: median %.1f  mean %.1f  p90 %.1f  p99 %.1f  max %.1f" %
      (np.median(hd), hd.mean(), np.percentile(hd, 90), np.percentile(hd, 99), hd.max()))
print("extent per axis (cells), median:", np.median(ext, axis=0))
