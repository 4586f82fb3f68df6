# Generator of random design scripts for the differential tests (tests/test_gpu_parity.py::test_random_scenes):
# `python random_tree.py SEED` is not how it is used -- the test imports make_script(seed) and runs the text it
# returns through `python -m designcsg_amd.run`.  Every helper of the design API that shapes the build procedure
# appears: leaves drawn and erased, capsules, hard and smooth unions / intersections (drawn and erased), nesting
# built by hand, axis-aligned objects at the origin (what the specialiser simplifies) next to rotated ones.
import numpy as np


def make_script(seed, big=False, smooth=True):
    """smooth=False: only what the reference's own API knows (no smooth joins), plus its remaining helpers -- draw_box,
    drawComponent / eraseComponent over a prefab with children, apply_transform -- for the front-end corpus
    (tests/golden/make_frontend_corpus.py runs these scripts on the reference Python)."""
    rng = np.random.default_rng(seed)
    L = ["from DesignCSG import *", "from designlibrary import *", "import numpy as np", "import scenecompiler", ""]
    brushes = ["sphere_brush", "box_brush", "cylinder_brush"]

    def tf():
        if rng.random() < 0.3:       # axis-aligned, sometimes at the origin, sometimes unit scale
            p = [0.0, 0.0, 0.0] if rng.random() < 0.5 else list(np.round(rng.uniform(-0.5, 0.5, 3), 3))
            s = [1.0, 1.0, 1.0] if rng.random() < 0.3 else list(np.round(rng.uniform(0.2, 0.9, 3), 3))
            return "Transform.initial(np.array(%r), 0.0, 0.0, 0.0, np.array(%r))" % (p, s)
        p = list(np.round(rng.uniform(-0.6, 0.6, 3), 3))
        a = list(np.round(rng.uniform(-3.0, 3.0, 3), 3))
        s = list(np.round(rng.uniform(0.15, 0.8, 3), 3))
        return "Transform.initial(np.array(%r), %r, %r, %r, np.array(%r))" % (p, a[0], a[1], a[2], s)

    def leaf():
        return "Component(%s, %s)" % (brushes[rng.integers(3)], tf())

    n_items = int(rng.integers(12, 20)) if big else int(rng.integers(3, 9))
    for _ in range(n_items):
        kind = rng.integers(0, 10)
        if kind <= 1:
            L.append("draw(%s, %s)" % (brushes[rng.integers(3)], tf()))
        elif kind == 2:
            L.append("erase(%s, %s)" % (brushes[rng.integers(3)], tf()))
        elif kind == 3:
            a, b = np.round(rng.uniform(-0.7, 0.7, 3), 3), np.round(rng.uniform(-0.7, 0.7, 3), 3)
            fn = "draw_capsule" if rng.random() < 0.6 else "cut_capsule"
            L.append("%s(np.array(%r), np.array(%r), T=%r)" % (fn, list(a), list(b), float(np.round(rng.uniform(0.05, 0.2), 3))))
        elif kind <= 6:
            fn = ["drawUnion", "eraseUnion", "drawIntersection", "eraseIntersection"][rng.integers(4)]
            kids = ", ".join(leaf() for _ in range(int(rng.integers(2, 10 if big else 5))))
            L.append("%s(%s, transform=%s)" % (fn, kids, tf()))
        elif kind <= 8 and not smooth:
            which = rng.integers(3)
            if which == 0:
                L.append("draw_box(np.array(%r), %r)" % (list(np.round(rng.uniform(-0.6, 0.6, 3), 3)), float(np.round(rng.uniform(0.1, 0.7), 3))))
            else:   # a prefab with children, instantiated under a transform (fabricate), once moved afterwards
                L.append("pf = %s" % leaf())
                for _ in range(int(rng.integers(1, 4))):
                    L.append("pf.add_child(%s)" % leaf())
                if rng.random() < 0.5:
                    L.append("pf.apply_transform(%s)" % tf())
                L.append("%s(pf, transform=%s)" % ("drawComponent" if which == 1 else "eraseComponent", tf()))
        elif kind <= 8:
            fn = ["drawSmoothUnion", "eraseSmoothUnion", "drawSmoothIntersection", "eraseSmoothIntersection"][rng.integers(4)]
            kids = ", ".join(leaf() for _ in range(int(rng.integers(2, 10 if big else 5))))
            L.append("%s(%s, k=%r, transform=%s)" % (fn, kids, float(np.round(rng.uniform(0.02, 0.3), 3)), tf()))
        else:   # hand-built nesting: a smooth or hard group holding a carved leaf and an inner group
            if smooth:
                outer = ("scenecompiler.SmoothComponent(%r, brush=compiler.null_brush(), material=compiler.default_material(), "
                         "transform=%s)" % (float(np.round(rng.uniform(0.03, 0.2), 3)), tf())) if rng.random() < 0.5 else \
                    ("scenecompiler.Component(brush=compiler.null_brush(), material=compiler.default_material(), transform=%s)" % tf())
            else:   # a hard group or an intersection group, drawn or carved
                outer = "scenecompiler.%s(brush=compiler.null_brush(), material=compiler.default_material(), transform=%s%s)" % (
                    "IntersectionComponent" if rng.random() < 0.4 else "Component", tf(), ", subtractive=True" if rng.random() < 0.3 else "")
            L.append("g = " + outer)
            L.append("g.add_child(%s)" % leaf())
            L.append("g.add_child(scenecompiler.Component(brush=%s, material=compiler.default_material(), transform=%s, "
                     "subtractive=True))" % (brushes[rng.integers(3)], tf()))
            L.append("h = scenecompiler.Component(brush=compiler.null_brush(), material=compiler.default_material(), transform=%s)" % tf())
            for _ in range(int(rng.integers(1, 4))):
                L.append("h.add_child(%s)" % leaf())
            L.append("g.add_child(h)")
            L.append("compiler.root.add_child(g)")
    L += ["", "setExportConfig(2.0, 3, 5, 6, np.pi / 4, 6, 16, 512, 4096)", "commit()", ""]
    return "\n".join(L)


if __name__ == "__main__":
    import sys
    print(make_script(int(sys.argv[1]) if len(sys.argv) > 1 else 0))
