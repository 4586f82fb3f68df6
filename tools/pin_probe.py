import os, sys, time, threading, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from designcsg_amd.evaluator import Evaluator
from conftest import read_scene, export_config_text
hip = C.CDLL("libamdhip64.so.7")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostFree.argtypes = [C.c_void_p]
s = read_scene("design2")
ev = Evaluator(0)
assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
cfg = export_config_text("design2", levels=[7, 9, 10], gd=50)
for _ in range(3): ev.export(cfg)            # no files: no pinned staging inside
def alloc(n, out):
    p = C.c_void_p(); t0 = time.perf_counter(); rc = hip.hipHostMalloc(C.byref(p), n, 0); out.append((rc, time.perf_counter() - t0)); hip.hipHostFree(p)
for size in (64 << 20, 144 << 20, 256 << 20):
    out = []; alloc(size, out); alone = out[0][1]
    t0 = time.perf_counter(); ev.export(cfg); t_export = time.perf_counter() - t0
    out = []; th = threading.Thread(target=alloc, args=(size, out)); t0 = time.perf_counter(); th.start(); ev.export(cfg); t_both_export = time.perf_counter() - t0; th.join(); t_both = time.perf_counter() - t0
    print("size %d MiB: hipHostMalloc alone %.2f ms; export alone %.2f ms; together: export %.2f ms, alloc %.2f ms, both done after %.2f ms" % (size >> 20, alone * 1e3, t_export * 1e3, t_both_export * 1e3, out[0][1] * 1e3, t_both * 1e3))
ev.close()
