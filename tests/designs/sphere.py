# Own design script (BASELINE config C2): one unit sphere brush at the origin.
# Written against the DesignCSG design API; runs unchanged on the reference
# Python modules (golden capture) and on designcsg_amd.frontend (parity test).
from DesignCSG import *
from designlibrary import *
import numpy as np

draw(sphere_brush, Transform.initial(
    position=np.array([0.0, 0.0, 0.0]), yaw=0.0, pitch=0.0, roll=0.0,
    scale=np.array([1.0, 1.0, 1.0])))

setExportConfig(2.0, 6, 8, 9, np.pi / 4, 50, 16, 512, 4096)
commit()
