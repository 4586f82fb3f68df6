"""Dense sweeps and an export of the 128-brush scene of the design's own brushes: threaded per-scene code against the stack
interpreter it replaces (DCSG_SPECIALISE_THREADED=0).  python tools/userbrush_times.py"""
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import bench
from designcsg_amd.evaluator import Evaluator
d = os.path.join("tests", "golden", "scenes", "userbrush128")
cfg = open(os.path.join(d, "exportConfig.txt")).read()
for tag, env in (("threaded", {}), ("interpreter", {"DCSG_SPECIALISE_THREADED": "0"})):
    os.environ.update(env)
    t0 = time.perf_counter()
    ev = Evaluator(0)
    assert ev.build_from_dir(d)[0] == 0
    t_load = time.perf_counter() - t0
    info = ev.scene_info()
    sw = bench.time_sweeps(ev, [0.0, 0.0, 0.0], 10.0, 512, reps=3, settle_s=0.05)
    exports = []
    for _ in range(3):
        t0 = time.perf_counter(); st = ev.export(cfg); exports.append(round(time.perf_counter() - t0, 4))
    print(json.dumps({"form": tag, "interpreter": info["interpreter"], "create_and_load_s": round(t_load, 3), "fused_ms": sw["fused"]["kernel_ms"],
                      "sdf_only_ms": sw["sdf_only"]["kernel_ms"], "export_s": exports, "triangles": int(st.triangles),
                      "stages_ms": dict(bbox=round(st.t_bbox * 1e3, 2), sweep=round(st.t_sweep * 1e3, 2), mesh=round(st.t_mesh * 1e3, 2), project=round(st.t_project * 1e3, 2))}))
    ev.close()
    for k in env: os.environ.pop(k)
