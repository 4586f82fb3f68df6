"""Diagnostic: milliseconds per 640x480 preview frame (dcsg_render_preview) for a few fixture scenes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from designcsg_amd.evaluator import Evaluator
ev = Evaluator(0)
for name in sys.argv[1:] or ["sphere", "design1", "design2", "mixed_groups", "flat256"]:
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes", name)
    assert ev.build_from_dir(d)[0] == 0
    t0 = time.perf_counter(); ev.render_preview(); first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(5):
        img = ev.render_preview()
    print("%-14s first frame (module build/load) %.3f s, then %.2f ms per frame; %d distinct colours"
          % (name, first, (time.perf_counter() - t0) / 5 * 1e3, len(set(map(tuple, img.reshape(-1, 3)[::7].tolist())))))
ev.close()
