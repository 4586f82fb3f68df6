import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from designcsg_amd.evaluator import Evaluator
from conftest import read_scene, export_config_text
for name in ("flat256", "design2"):
    s = read_scene(name)
    ev = Evaluator(0)
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
    cfg = export_config_text(name, levels=[7, 9, 10], gd=50)
    for it in range(2):
        t0 = time.perf_counter(); st = ev.export(cfg); t1 = time.perf_counter()
        print(name, "whole", round(t1 - t0, 4), dict(sweep=round(st.t_sweep*1e3,2), mesh=round(st.t_mesh*1e3,2), proj=round(st.t_project*1e3,2)))
    for world in (4, 8):
        for it in range(2):
            out = []
            for r in range(world):
                t0 = time.perf_counter(); st = ev.export_slab(cfg, r, world); t1 = time.perf_counter()
                out.append((round(t1 - t0, 4), round(st.t_bbox*1e3,2), round(st.t_sweep*1e3,2), round(st.t_mesh*1e3,2), round(st.t_project*1e3,2), int(st.triangles)))
            print(name, "world", world, "iter", it, out)
    ev.close()
