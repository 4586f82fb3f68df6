import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from designcsg_amd.evaluator import Evaluator
from conftest import read_scene, export_config_text
for name in ("flat256", "grouped256", "smooth256", "design2"):
    s = read_scene(name)
    ev = Evaluator(0)
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
    st = ev.export(export_config_text(name, levels=[7, 9, 10], gd=50))
    m = st.mesh
    print(name, "swept_nodes", st.swept_nodes, "swept_points", st.swept_points, "sweep ms", round(st.sweep_kernel_ms, 3))
    for l in range(2, 10):
        print("  level", l, "nodes", m.nodes[l], "culled", m.culled[l], "subdivided", m.subdivided[l], "leaves", m.leaves[l])
    ev.close()
