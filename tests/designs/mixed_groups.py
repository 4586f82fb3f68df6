# Own design script exercising every DesignCSG helper that changes the build
# procedure: capsules (nested fabricate), union / intersection groups, and the
# subtractive variants of each.
from DesignCSG import *
from designlibrary import *
import numpy as np


def tf(px, py, pz, yaw=0.0, pitch=0.0, roll=0.0, s=(1.0, 1.0, 1.0)):
    return Transform.initial(position=np.array([px, py, pz], dtype=float), yaw=yaw,
                             pitch=pitch, roll=roll, scale=np.array(s, dtype=float))


draw_box(np.array([0.0, -0.2, 0.0]), 0.9)
draw_capsule(np.array([-0.6, 0.1, 0.0]), np.array([0.6, 0.4, 0.2]), T=0.15)
cut_capsule(np.array([0.0, -0.7, -0.1]), np.array([0.0, 0.7, 0.1]), T=0.12)

drawUnion(Component(sphere_brush, tf(0.3, 0.3, 0.0, s=(0.4, 0.4, 0.4))),
          Component(cylinder_brush, tf(-0.3, 0.3, 0.0, roll=0.5, s=(0.2, 0.6, 0.2))),
          transform=tf(0.0, 0.1, 0.3, yaw=0.3))

drawIntersection(Component(sphere_brush, tf(0.0, 0.0, 0.0, s=(0.7, 0.7, 0.7))),
                 Component(box_brush, tf(0.1, 0.0, 0.0, pitch=0.4, s=(0.5, 0.5, 0.5))),
                 transform=tf(0.0, 0.55, -0.3))

eraseUnion(Component(sphere_brush, tf(0.45, -0.2, 0.45, s=(0.3, 0.3, 0.3))),
           Component(sphere_brush, tf(-0.45, -0.2, 0.45, s=(0.3, 0.3, 0.3))))

eraseIntersection(Component(box_brush, tf(0.0, -0.2, -0.45, s=(0.5, 0.3, 0.3))),
                  Component(cylinder_brush, tf(0.0, -0.2, -0.45, s=(0.3, 0.8, 0.3))))

erase(sphere_brush, tf(0.0, -0.65, 0.0, s=(0.35, 0.2, 0.35)))

setExportConfig(2.0, 4, 6, 7, np.pi / 4, 20, 16, 512, 4096)
commit()
