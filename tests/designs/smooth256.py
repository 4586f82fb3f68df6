# Own design script (BASELINE config 5 as worded: a deep CSG tree of >= 256 primitives with smooth-min
# unions).  EXTENSION opcodes 6-8 -- not readable by the reference.  16 smooth groups of 16 leaves; every
# fourth group is carved out of the rest.
from DesignCSG import *
from designlibrary import *
import numpy as np

rng = np.random.default_rng(97531)
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
    if g % 4 == 3:
        eraseSmoothUnion(*leaves, k=0.05, transform=gt)
    else:
        drawSmoothUnion(*leaves, k=0.1, transform=gt)

setExportConfig(2.0, 7, 9, 10, np.pi / 4, 50, 16, 512, 4096)
commit()
