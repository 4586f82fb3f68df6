"""Diagnostic: fused sweep time of Z slabs of different heights of the 513^3 sphere lattice, 20 launches each after a
2-launch lead-in, in one process (run from the repo root on the GPU box).  The first configuration is measured on a GPU
that has just been idle and is the slow one whatever its size; the list therefore ends with the sizes it began with.
This is what exposed the clock-settling effect described in DESIGN.md section 6."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.getcwd())
import torch
from designcsg_amd import _lib
from designcsg_amd.evaluator import Evaluator
d = "tests/golden/scenes/sphere"
ev = Evaluator(0); st, log = ev.build_from_dir(d); assert st == 0
center, diameter, _ = ev.estimate_bounding_box(2.0)
L, h = _lib.lib(), ev._h
for res in (512,):
    n1 = res + 1
    sdf = torch.empty(n1 ** 3, dtype=torch.float32, device="cuda"); nrm = torch.empty(3 * n1 ** 3, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()
    c3, d3, sp = (C.c_float * 3)(*center), (C.c_float * 3)(diameter, diameter, diameter), C.c_void_p(stream.cuda_stream)
    for planes in (513, 512, 511, 480, 448, 416, 384, 352, 320, 288, 257, 256, 255, 224, 192, 160, 128, 513, 256):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        for i in range(reps + 2):
            if i == 2: e0.record(stream)
            rc = L.dcsg_sweep_lattice_device(h, c3, d3, res, 0, planes, C.c_void_p(sdf.data_ptr()), C.c_void_p(nrm.data_ptr()), sp)
            assert rc == 0
        e1.record(stream); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        pts = planes * n1 * n1
        print("res %d planes %4d: %.4f ms  %.3f ns per 1000 points  %.2f TB/s" % (res, planes, ms, ms * 1e6 / pts * 1000, 16 * pts / ms / 1e9))
    del sdf, nrm
