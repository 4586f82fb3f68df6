"""Diagnostic: achievable HBM write bandwidth on this box for plain fills (torch fill_ / zero_), at the byte
counts of the SDF-only and fused sweeps.  python tools/write_bw_probe.py"""
import torch

def t(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for n in (135005697, 4 * 135005697, 1076890625):
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    ms = t(lambda: x.fill_(1.5))
    print("fill_ %d floats: %.4f ms  %.0f GB/s" % (n, ms, 4 * n / ms / 1e6))
    ms = t(lambda: x.zero_())
    print("zero_ %d floats: %.4f ms  %.0f GB/s" % (n, ms, 4 * n / ms / 1e6))
    y = torch.empty_like(x)
    ms = t(lambda: y.copy_(x))
    print("copy_ %d floats: %.4f ms  %.0f GB/s (read+write)" % (n, ms, 8 * n / ms / 1e6))
    del x, y
