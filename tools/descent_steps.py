#!/usr/bin/env python3
"""How many of performGradientDescent's steps does a vertex need?  Exports a scene with 0 projection steps, then projects the
distinct vertices with 1, 2, ... 50 steps (torch tensor, dcsg_project_vertices_device) and finds for every vertex the step at
which the kernel retires it: the first k with p_(k+1) == p_k (fixed point) or p_(k+1) == p_(k-1) (period two).  Prints the
histogram and what it means for a wave of 64 consecutive vertices (a wave runs until its slowest vertex retires)."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from designcsg_amd.evaluator import Evaluator
from conftest import read_scene, export_config_text

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="design2")
ap.add_argument("--levels", type=int, nargs=3, default=[6, 8, 9])
a = ap.parse_args()
s = read_scene(a.scene)
ev = Evaluator(0)
assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
ev.export(export_config_text(a.scene, levels=a.levels, gd=0))
tris = ev.download_triangles().reshape(-1, 3)
# distinct vertices in order of first appearance (what the export projects)
_, first = np.unique(tris.view(np.uint32).reshape(-1, 3), axis=0, return_index=True)
verts = tris[np.sort(first)]
n = len(verts)
steps = 50
P = [verts.copy()]
for k in range(1, steps + 1):
    t = torch.from_numpy(verts.copy()).cuda()
    ev.project_vertices_device(t.data_ptr(), n, k)
    torch.cuda.synchronize()
    P.append(t.cpu().numpy())
B = [p.view(np.uint32) for p in P]
retire = np.full(n, steps, dtype=np.int64)     # number of evaluations the kernel runs for the vertex
alive = np.ones(n, dtype=bool)
for k in range(steps):   # evaluation k computes q = F(p_k)
    # NOTE: P[k+1] from a run with k+1 steps equals p_(k+1) only while the vertex is alive (a retired vertex reports its final position)
    fixed = (B[k + 1] == B[k]).all(1)
    period = (B[k + 1] == B[k - 1]).all(1) if k >= 1 else np.zeros(n, bool)
    done = alive & (fixed | period)
    retire[done] = k + 1
    alive &= ~done
hist = np.bincount(retire, minlength=steps + 1)
print("%s levels %s: %d distinct vertices; evaluations per vertex: mean %.1f, median %d; never retired (all %d): %.1f %%" % (
    a.scene, a.levels, n, retire.mean(), np.median(retire), steps, 100.0 * (retire == steps).mean()))
print("histogram (evaluations: vertices):", " ".join("%d:%d" % (k, c) for k, c in enumerate(hist) if c))
w = retire[: n // 64 * 64].reshape(-1, 64)
print("a wave of 64 consecutive vertices runs max over its lanes: mean %.1f evaluations; lane-evaluations used / issued = %.2f" % (
    w.max(1).mean(), w.sum() / (w.max(1).sum() * 64.0)))
for cut in (4, 8, 12, 16):
    left = (retire > cut).mean()
    print("  after %2d evaluations %.1f %% of the vertices are still moving" % (cut, 100 * left))
ev.close()
