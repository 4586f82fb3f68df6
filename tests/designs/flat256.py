# Own design script (BASELINE config C5, flat variant): 256 built-in brushes
# unioned at the root.  Parameters follow SURVEY.md section 8(d) exactly.
from DesignCSG import *
from designlibrary import *
import numpy as np

rng = np.random.default_rng(12345)
brushes = [sphere_brush, box_brush, cylinder_brush]
for i in range(256):
    p = rng.uniform(-0.8, 0.8, 3)
    s = rng.uniform(0.05, 0.2, 3)
    a = rng.uniform(0, np.pi, 3)
    draw(brushes[i % 3], Transform.initial(p, a[0], a[1], a[2], s))

setExportConfig(2.0, 7, 9, 10, np.pi / 4, 50, 16, 512, 4096)
commit()
