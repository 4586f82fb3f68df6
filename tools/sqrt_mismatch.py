"""Diagnostic: where does the kernels' square root differ from IEEE sqrtf?  (all 2^32 inputs, builtin module)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from designcsg_amd.evaluator import Evaluator
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes", "flat256")
ev = Evaluator(0)
assert ev.build(open(d + "/scene.txt").read(), open(d + "/buildprocedure.txt").read(), None)[0] == 0
chunk = 1 << 26
out = torch.empty(chunk, dtype=torch.float32, device="cuda")
total = 0
for c in range(0, 1 << 32, chunk):
    x = (torch.arange(c, c + chunk, dtype=torch.int64, device="cuda") - (1 << 32) * (c >= (1 << 31))).to(torch.int32).view(torch.float32)
    ev.debug_sqrt_device(x.data_ptr(), chunk, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = torch.sqrt(x.double()).float()
    bad = ~((want.view(torch.int32) == out.view(torch.int32)) | (torch.isnan(want) & torch.isnan(out)))
    n = int(bad.sum())
    if n:
        i = int(torch.nonzero(bad)[0])
        print("chunk %#010x: %d mismatches, first x=%#010x (%r) got %r want %r" % (c, n, c + i, float(x[i]), float(out[i]), float(want[i])))
        total += n
print("total mismatches", total)
