# Own design script for the smooth-join EXTENSION (opcodes 6-8; not readable by the reference): smooth unions
# with different blend radii, a subtractive smooth group, a smooth intersection, a subtractive leaf inside a
# smooth group and a nested hard group inside a smooth one, mixed with plain draw / erase.
from DesignCSG import *
from designlibrary import *
import numpy as np
import scenecompiler


def tf(px, py, pz, yaw=0.0, pitch=0.0, roll=0.0, s=(1.0, 1.0, 1.0)):
    return Transform.initial(position=np.array([px, py, pz], dtype=float), yaw=yaw,
                             pitch=pitch, roll=roll, scale=np.array(s, dtype=float))


draw_box(np.array([0.0, -0.45, 0.0]), 0.7)

drawSmoothUnion(Component(sphere_brush, tf(-0.25, 0.0, 0.0, s=(0.5, 0.5, 0.5))),
                Component(sphere_brush, tf(0.2, 0.1, 0.05, s=(0.4, 0.4, 0.4))),
                Component(box_brush, tf(0.0, 0.25, -0.1, yaw=0.4, s=(0.3, 0.3, 0.3))),
                Component(cylinder_brush, tf(0.0, -0.1, 0.25, roll=0.7, s=(0.15, 0.6, 0.15))),
                k=0.15, transform=tf(0.0, 0.1, 0.0, yaw=0.2))

eraseSmoothUnion(Component(sphere_brush, tf(0.35, 0.3, 0.3, s=(0.3, 0.3, 0.3))),
                 Component(sphere_brush, tf(0.15, 0.35, 0.35, s=(0.25, 0.25, 0.25))),
                 k=0.05)

drawSmoothIntersection(Component(sphere_brush, tf(0.0, 0.0, 0.0, s=(0.6, 0.6, 0.6))),
                       Component(box_brush, tf(0.05, 0.0, 0.0, pitch=0.3, s=(0.45, 0.45, 0.45))),
                       k=0.08, transform=tf(-0.55, 0.35, -0.3))

# a smooth group built by hand: a carved leaf and a nested hard union inside it
grp = scenecompiler.SmoothComponent(0.12, brush=compiler.null_brush(), material=compiler.default_material(),
                                    transform=tf(0.5, -0.1, -0.35))
grp.add_child(Component(sphere_brush, tf(0.0, 0.0, 0.0, s=(0.45, 0.45, 0.45))))
grp.add_child(scenecompiler.Component(brush=box_brush, material=compiler.default_material(),
                                      transform=tf(0.1, 0.15, 0.0, s=(0.3, 0.3, 0.3)), subtractive=True))
inner = scenecompiler.Component(brush=compiler.null_brush(), material=compiler.default_material(),
                                transform=tf(-0.15, -0.1, 0.1))
inner.add_child(Component(cylinder_brush, tf(0.0, 0.0, 0.0, s=(0.2, 0.5, 0.2))))
inner.add_child(Component(sphere_brush, tf(0.0, 0.25, 0.0, s=(0.25, 0.25, 0.25))))
grp.add_child(inner)
compiler.root.add_child(grp)

erase(sphere_brush, tf(0.0, -0.75, 0.0, s=(0.4, 0.25, 0.4)))

setExportConfig(2.0, 4, 6, 7, np.pi / 4, 20, 16, 512, 4096)
commit()
