#!/usr/bin/env python3
"""Diagnostic: what back-to-back launches of the SDF-only sweep cost beyond the kernel itself (host time per
dcsg_sweep_lattice_device call, device time per launch over a region of launches).  python tools/launch_gap.py"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from designcsg_amd import _lib  # noqa: E402
from designcsg_amd.evaluator import Evaluator  # noqa: E402

ev = Evaluator(0)
assert ev.build_from_dir(os.path.join(ROOT, "tests", "golden", "scenes", "sphere"))[0] == 0
res, n1 = 512, 513
sdf = torch.empty(n1 ** 3, dtype=torch.float32, device="cuda")
nrm = torch.empty(3 * n1 ** 3, dtype=torch.float32, device="cuda")
L, h = _lib.lib(), ev._h
c3, d3 = (C.c_float * 3)(0, 0, 0), (C.c_float * 3)(5.3515625, 5.3515625, 5.3515625)
stream = torch.cuda.current_stream()
sp = C.c_void_p(stream.cuda_stream)
out = {}
for key, nptr in (("sdf_only", None), ("fused", nrm.data_ptr())):
    call = lambda: L.dcsg_sweep_lattice_device(h, c3, d3, res, 0, n1, C.c_void_p(sdf.data_ptr()), C.c_void_p(nptr) if nptr else None, sp)
    for _ in range(600):
        call()
    torch.cuda.synchronize()
    reps = 200
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    t_host = time.perf_counter() - t0
    b.record(stream)
    torch.cuda.synchronize()
    out[key] = {"host_us_per_call": round(t_host / reps * 1e6, 2), "device_us_per_launch_over_region": round(a.elapsed_time(b) / reps * 1e3, 2)}
print(json.dumps(out))
ev.close()
