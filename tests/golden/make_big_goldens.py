#!/usr/bin/env python3
"""Runs the CPU oracle on the full-size configurations (BASELINE C2/C3/C4/C5) and records counts and
order-independent checksums in tests/golden/big_configs.json.  Minutes of CPU; run once per oracle change.
The C2/C3/C4 figures must equal SURVEY.md 8(c) (tests/test_oracle_golden.py checks that)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

TABLE = O.LookupTable(open(os.path.join(HERE, "..", "..", "designcsg_amd", "data", "lookupTable.txt")).read())
RUNS = {"sphere_6_8_9": ("sphere", (6, 8, 9)), "design2_6_8_9": ("design2", (6, 8, 9)),
        "design2_7_9_10": ("design2", (7, 9, 10)), "flat256_6_8_9": ("flat256", (6, 8, 9)),
        "grouped256_6_8_9": ("grouped256", (6, 8, 9)),
        "smooth256_6_8_9": ("smooth256", (6, 8, 9)),   # smooth-join extension: oracle-defined, no reference figure
        # BASELINE config 5 at its stated size (1024^3, levels 7/9/10)
        "flat256_7_9_10": ("flat256", (7, 9, 10)), "grouped256_7_9_10": ("grouped256", (7, 9, 10)),
        "smooth256_7_9_10": ("smooth256", (7, 9, 10))}   # extension, as above


def tri_set_hash(tris):
    b = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9).view(np.uint32)
    b = b[np.lexsort(b.T[::-1])]
    return hashlib.sha256(b.tobytes()).hexdigest()


def main():
    """No arguments: all runs.  With arguments: only those keys, merged into the existing file."""
    path = os.path.join(HERE, "big_configs.json")
    only = sys.argv[1:]
    out = json.load(open(path)) if only else {}
    for key, (name, levels) in RUNS.items():
        if only and key not in only:
            continue
        d = os.path.join(HERE, "scenes", name)
        sc = O.Scene.from_dir(d)
        cfg = O.parse_export_config(open(os.path.join(d, "exportConfig.txt")).read())
        cfg.min_level, cfg.max_level, cfg.grid_level = levels
        c, dia, n = sc.estimate_bounding_box(cfg.bounding_box_half_diameter)
        tris, st = sc.get_surface(c, dia[0], cfg, TABLE)
        gd = sc.gradient_descent(tris, 50)
        v = gd.reshape(-1, 3).astype(np.float64)
        out[key] = {
            "scene": name, "levels": list(levels), "box_center": c, "box_diameter": dia[0], "interior_points": n,
            "triangles": int(len(tris)),
            "per_level": {str(l): int(st.triangles[l]) for l in range(32) if st.triangles[l]},
            "culled": {str(l): int(st.nodes_culled[l]) for l in range(32) if st.nodes_culled[l]},
            "subdivided": {str(l): int(st.nodes_subdivided[l]) for l in range(32) if st.nodes_subdivided[l]},
            "pre_gd_set_sha256": tri_set_hash(tris), "gd50_set_sha256": tri_set_hash(gd),
            "gd50_sums": v.sum(0).tolist(), "gd50_sum_abs": float(np.linalg.norm(v, axis=1).sum()),
            "gd50_nan": int(np.isnan(v).sum()),
        }
        print(key, out[key]["triangles"], out[key]["per_level"], flush=True)
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
