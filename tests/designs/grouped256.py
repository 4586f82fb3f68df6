# Own design script (BASELINE config C5, grouped variant): 16 groups of 16
# leaves, cycling union / intersection / subtractive-union joins, so that the
# interpreter uses 17 stack slots plus R0.
from DesignCSG import *
from designlibrary import *
import numpy as np

rng = np.random.default_rng(54321)
brushes = [sphere_brush, box_brush, cylinder_brush]
for g in range(16):
    centre = rng.uniform(-0.6, 0.6, 3)
    leaves = []
    for i in range(16):
        p = rng.uniform(-0.25, 0.25, 3)
        s = rng.uniform(0.1, 0.3, 3)
        a = rng.uniform(0, np.pi, 3)
        leaves.append(Component(brushes[(g + i) % 3], Transform.initial(p, a[0], a[1], a[2], s)))
    gt = Transform.initial(centre, 0.0, 0.0, 0.0, np.array([1.0, 1.0, 1.0]))
    if g % 3 == 0:
        drawUnion(*leaves, transform=gt)
    elif g % 3 == 1:
        drawIntersection(*leaves, transform=gt)
    else:
        eraseUnion(*leaves, transform=gt)

setExportConfig(2.0, 7, 9, 10, np.pi / 4, 50, 16, 512, 4096)
commit()
