"""Per-rank stage times of the Z-slab export on ONE GPU (the ranks one after the other; on N GPUs they run concurrently, so the
slowest rank is the sharded compute time): equal slabs against the work-balanced cut planes.
    python tools/slab_times.py [scene ...] > gpurun_out/slab_times.txt"""
import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from designcsg_amd.evaluator import Evaluator
from conftest import read_scene, export_config_text
for name in (sys.argv[1:] or ("flat256", "design2")):
    s = read_scene(name)
    ev = Evaluator(0)
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
    cfg = export_config_text(name, levels=[7, 9, 10], gd=50)
    whole = []
    for it in range(3):
        t0 = time.perf_counter(); st = ev.export(cfg); t1 = time.perf_counter()
        whole.append(t1 - t0)
        print(name, "whole", round(t1 - t0, 4), dict(bbox=round(st.t_bbox*1e3,2), sweep=round(st.t_sweep*1e3,2), mesh=round(st.t_mesh*1e3,2), proj=round(st.t_project*1e3,2)))
    for world in (2, 4, 8):
        for equal in (True, False):
            best = None
            for it in range(3):
                out = []
                for r in range(world):
                    t0 = time.perf_counter(); st = ev.export_slab(cfg, r, world, equal_slabs=equal); t1 = time.perf_counter()
                    out.append((round((t1 - t0) * 1e3, 2), int(st.triangles), (st.slab_z_lo, st.slab_z_hi),
                                "A %.2f sweep %.2f mesh %.2f proj %.2f" % (st.t_bbox * 1e3, st.t_sweep * 1e3, st.t_mesh * 1e3, st.t_project * 1e3)))
                if best is None or max(o[0] for o in out) < max(o[0] for o in best):
                    best = out
            print(name, "world", world, "equal" if equal else "balanced", "slowest rank %.2f ms of whole %.2f ms (x%.2f)" % (
                max(o[0] for o in best), min(whole) * 1e3, min(whole) * 1e3 / max(o[0] for o in best)), "per rank (ms, triangles, planes, stages):", best)
    ev.close()
