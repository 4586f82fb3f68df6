# Own design script: prefab components (children under a parent, fabricate, apply_transform), instanced with
# drawComponent / eraseComponent, next to drawBrush / eraseBrush aliases.
from DesignCSG import *
from designlibrary import *
import numpy as np


def tf(p, yaw=0.0, pitch=0.0, roll=0.0, s=(1.0, 1.0, 1.0)):
    return Transform.initial(np.array(p, dtype=float), yaw, pitch, roll, np.array(s, dtype=float))


peg = Component(cylinder_brush, tf([0.0, 0.0, 0.0], s=(0.15, 0.5, 0.15)))
peg.add_child(Component(sphere_brush, tf([0.0, 0.5, 0.0], s=(1.6, 0.5, 1.6))))
peg.apply_transform(tf([0.0, 0.1, 0.0], roll=0.2))

drawBrush(box_brush, tf([0.0, -0.3, 0.0], s=(1.2, 0.2, 1.2)))
for k in range(4):
    a = k * np.pi / 2
    drawComponent(peg, transform=tf([0.4 * np.cos(a), 0.0, 0.4 * np.sin(a)], yaw=a))
eraseComponent(peg, transform=tf([0.0, -0.2, 0.0], s=(0.6, 1.0, 0.6)))
eraseBrush(sphere_brush, tf([0.0, -0.45, 0.0], s=(0.4, 0.2, 0.4)))

setExportConfig(2.0, 4, 6, 7, np.pi / 4, 10, 16, 512, 4096)
commit()
