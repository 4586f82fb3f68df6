"""Design-script helpers -- own implementation of the reference's user-facing module.

    from DesignCSG import *
    draw(sphere_brush, Transform.initial(position, yaw, pitch, roll, scale))
    setExportConfig(2.0, 6, 8, 9, np.pi/4, 50)
    commit()

Defines the three standard brushes (ids 2, 3, 4 after the compiler's null and void brushes), the
draw / erase family, capsules, grouped unions / intersections and `setExportConfig`, which writes
exportConfig.txt (nine lines, parsed positionally by dcsg_export; format of DesignCSG.py:205-237 of the
reference).  Outputs are pinned byte for byte by tests/test_frontend.py.
"""
import numpy as np

import scenecompiler

compiler = scenecompiler.SceneCompiler()

# Brush bodies are OpenCL C; `v` is the point in the brush's local frame.  Unit shapes: radius / half
# extent 0.5, sized by the component transform.
sphere_brush = compiler.define_brush(body="return length(v)-0.5;")
cylinder_brush = compiler.define_brush(body="""

    v=fabs(v);
    float x = length((float2)(v.x,v.z));
    float y = v.y;
    return max(x-0.5,y-0.5);

""")

box_brush = compiler.define_brush(body="""
    v=fabs(v);
    return max(v.x-0.5,max(v.y-0.5,v.z-0.5));
""")

define_brush = compiler.define_brush
define_material = compiler.define_material
addArbitraryData = compiler.addArbitraryData
commit = compiler.commit
define_auxillary_function = compiler.define_auxillary_function
add_preprocessor_define = compiler.add_preprocessor_define
Transform = scenecompiler.Transform
PI = np.pi


def _node(brush, tf, subtractive=False):
    return scenecompiler.Component(brush=brush, material=compiler.default_material(), transform=tf,
                                   subtractive=subtractive)


def draw(brush, tf):
    compiler.root.add_child(_node(brush, tf))


def erase(brush, tf):
    compiler.root.add_child(_node(brush, tf, subtractive=True))


drawBrush = draw
eraseBrush = erase


def Component(brush, transform=Transform.identity()):
    return scenecompiler.Component(brush=brush, material=compiler.default_material(), transform=transform)


def _capsule(A, B, T, subtractive):
    """Cylinder of thickness T from A to B with a sphere cap at each end, as one prefab instance."""
    axis = B - A
    length = np.linalg.norm(axis)
    unscaled = lambda p, s: scenecompiler.Transform.initial(position=np.array(p), yaw=0, pitch=0, roll=0,
                                                             scale=np.array(s))
    body = _node(cylinder_brush, unscaled([0.0, 0.0, 0.0], [T, length, T]))
    for end in (0.5, -0.5):
        # caps undo the body's stretch along its axis so that they stay round
        body.add_child(_node(sphere_brush, unscaled([0.0, end, 0.0], [1, T / length, 1])))
    direction = scenecompiler.Transform.normalized(axis)
    heading = np.arctan2(direction[2], direction[0])
    elevation = np.arcsin(direction[1])
    placement = scenecompiler.Transform.initial(position=(A + B) / 2, yaw=np.pi / 2 - heading,
                                                pitch=elevation - np.pi / 2, roll=0,
                                                scale=np.array([1.0, 1.0, 1.0]))
    if subtractive:
        compiler.root.add_child(body.fabricate(transform=placement, subtractive=True))
    else:
        compiler.root.add_child(body.fabricate(transform=placement))


def draw_capsule(A, B, T=1):
    _capsule(A, B, T, False)


def cut_capsule(A, B, T=1):
    _capsule(A, B, T, True)


def draw_box(origin, diameter):
    compiler.root.add_child(_node(box_brush, scenecompiler.Transform.initial(
        position=origin, yaw=0, pitch=0, roll=0, scale=diameter * np.ones((3,), dtype=float))))


def drawComponent(component, transform=Transform.identity()):
    compiler.root.add_child(component.fabricate(transform=transform))


def eraseComponent(component, transform=Transform.identity()):
    compiler.root.add_child(component.fabricate(transform=transform, subtractive=True))


def _group(kind, components, transform, subtractive):
    kw = dict(brush=compiler.null_brush(), material=compiler.default_material(), transform=transform)
    if subtractive:
        kw["subtractive"] = True
    group = kind(**kw)
    for component in components:
        group.add_child(component)
    compiler.root.add_child(group)


def drawUnion(*components, transform=Transform.identity()):
    _group(scenecompiler.Component, components, transform, False)


def eraseUnion(*components, transform=Transform.identity()):
    _group(scenecompiler.Component, components, transform, True)


def drawIntersection(*components, transform=Transform.identity()):
    _group(scenecompiler.IntersectionComponent, components, transform, False)


def eraseIntersection(*components, transform=Transform.identity()):
    _group(scenecompiler.IntersectionComponent, components, transform, True)


# ---- extension of this package (not in the reference): smooth groups ------------------------------------
# The children of the group are blended with the polynomial smooth minimum / maximum of radius k
# (build-procedure opcodes 6-8); scenes using them are not readable by the reference.
def _smooth_group(k, intersection, components, transform, subtractive):
    kind = lambda **kw: scenecompiler.SmoothComponent(k, intersection, **kw)
    _group(kind, components, transform, subtractive)


def drawSmoothUnion(*components, k=0.1, transform=Transform.identity()):
    _smooth_group(k, False, components, transform, False)


def eraseSmoothUnion(*components, k=0.1, transform=Transform.identity()):
    _smooth_group(k, False, components, transform, True)


def drawSmoothIntersection(*components, k=0.1, transform=Transform.identity()):
    _smooth_group(k, True, components, transform, False)


def eraseSmoothIntersection(*components, k=0.1, transform=Transform.identity()):
    _smooth_group(k, True, components, transform, True)


def setExportConfig(boundingBoxHalfDiameter, minimumOctreeLevel, maximumOctreeLevel, gridLevel,
                    complexSurfaceThreshold, gradientDescentSteps=10, cacheSubdivision=16, queriesBeforeGC=64,
                    queriesBeforeFree=1024, meshSubdivisionLevel=4, maxPoolSize=0):
    """exportConfig.txt: nine values, one per line.  The first is in world units, i.e. already multiplied
    by the root scale of 5; meshSubdivisionLevel / maxPoolSize are accepted and ignored, as in the reference."""
    values = [5.0 * boundingBoxHalfDiameter, minimumOctreeLevel, maximumOctreeLevel, gridLevel,
              complexSurfaceThreshold, gradientDescentSteps, cacheSubdivision, queriesBeforeGC, queriesBeforeFree]
    with open("exportConfig.txt", "w") as fl:
        fl.write("".join(str(v) + "\n" for v in values))
