#!/usr/bin/env python3
"""Golden vectors of the REFERENCE's own kernel `k2`, run on an MI355X.

Run on a GPU box (`gpurun -- python tests/golden/make_k2_vectors.py`): loads the code objects that
`make -C oracle ref-k2` built in the build container from /root/reference/master/k2.cl (+ the fixture's scene.cl,
i.e. k2build.cl) into oracle/_ref/ and launches them as the reference's Evaluator does (12 buffers, local size 1,
oracle/ref_k2.py).  Writes gpurun_out/k2_vectors.npz; the committed copy is tests/golden/k2_vectors.npz.  The file
holds OUTPUTS of the reference kernel and the integer seeds / indices that regenerate its inputs -- no reference text.

Per scene (design2, design1, sphere, mixed_groups, flat256) and flavour (`strict` = reference options + IEEE basic
operations, stored as f32; `default` = the reference's own options, stored as the signed ulp distance from `strict`):
  * SDF and normal at 100 000 seeded points: 60 000 uniform in [-5,5]^3, 40 000 near the surface (the 40 000 of 4 000 000
    further uniform points with the smallest |sdf|; their indices are stored);
  * SDF on the 64^3 lattice over [-5,5)^3 (Design2 has 6 089 negative samples there);
plus, for Design2, the survey's probe points; stage A (DesignCSG.cpp:668-712) over the reference kernel for sphere /
design1 / design2; and cms::performGradientDescent (mesh.hpp:531-593, 50 steps) over the reference kernel on the
vertices of the 4/6/7 mesh of sphere and design2 (input: the oracle's raw triangles in sorted order).
"""
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_k2 as R  # noqa: E402

VECTOR_SCENES = ("design2", "design1", "sphere", "mixed_groups", "flat256")
N_UNIFORM, N_NEAR, N_POOL = 60000, 40000, 4000000
STAGE_A_SCENES = ("sphere", "design1", "design2")
GD_SCENES = ("sphere", "design2")
GD_LEVELS, GD_STEPS = (4, 6, 7), 50
PROBE_SDF = [(0, 0, 0), (1, 0, 0), (0, -4.3, 0), (2.5, 2.5, 2.5), (0.3, 0.1, -0.7)]
PROBE_NORMAL = [(1, 0, 0), (0, -4.3, 0), (2.5, 2.5, 2.5)]


def scene_seed(name):
    return int.from_bytes(hashlib.sha256(("k2-vectors:" + name).encode()).digest()[:4], "little")


def uniform_points(name):
    return np.random.default_rng(scene_seed(name)).uniform(-5, 5, (N_UNIFORM, 3)).astype(np.float32)


def pool_points(name):
    return np.random.default_rng(scene_seed(name) + 1).uniform(-5, 5, (N_POOL, 3)).astype(np.float32)


def lattice_points():
    xs = (-5 + 10 * np.arange(64) / 64).astype(np.float32)
    return np.stack(np.meshgrid(xs, xs, xs, indexing="ij"), -1).reshape(-1, 3)


def ulp_distance(a, b):
    """Signed distance b - a in units of float32 ordering (monotone integer map of the bit patterns)."""
    return _key(b) - _key(a)


def _key(x):
    i = np.ascontiguousarray(x, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.where(i < 0, -(i & 0x7FFFFFFF), i)


def apply_delta(strict, small):
    """Inverse of pack_delta without the exceptions: the float `small` ulps away from `strict`."""
    k = _key(strict).reshape(-1) + small.astype(np.int64)
    b = np.where(k < 0, (-k) | 0x80000000, k).astype(np.uint32)
    return b.view(np.float32)


def pack_delta(strict, other):
    """`other` as int8 ulp distances from `strict`, plus the exceptions (distance too large, -0.0, NaN payloads)."""
    o = np.ascontiguousarray(other, dtype=np.float32).reshape(-1)
    strict = np.ascontiguousarray(strict, dtype=np.float32).reshape(-1)
    small = np.clip(ulp_distance(strict, o), -127, 127).astype(np.int8)
    big = np.flatnonzero(apply_delta(strict, small).view(np.uint32) != o.view(np.uint32))
    return small, big.astype(np.int64), o[big]


def unpack_delta(strict, small, big_index, big_value):
    o = apply_delta(strict, small).copy()
    o[big_index] = big_value
    return o.reshape(np.shape(strict))


def load_vectors(path=None):
    """(npz, meta) of the committed vectors."""
    z = np.load(path or os.path.join(HERE, "k2_vectors.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def vector(z, scene, flavour, key):
    strict = z["%s/strict/%s" % (scene, key)]
    if flavour == "strict":
        return strict
    p = "%s/default/%s" % (scene, key)
    return unpack_delta(strict, z[p + ".d8"], z[p + ".big_index"], z[p + ".big_value"])


def scene_points(z, scene):
    """The 100 000 evaluation points of a scene, regenerated from the seeds and the stored near-surface indices."""
    return np.concatenate([uniform_points(scene), pool_points(scene)[z[scene + "/near_index"]]])


def sorted_triangles(tris):
    b = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
    return b[np.lexsort(b.view(np.uint32).T[::-1])].reshape(-1, 3, 3)


def raw_mesh_vertices(name):
    """Input of the GD pin: the oracle's raw 4/6/7 triangles (deterministic as a set), sorted."""
    from oracle import oracle as O
    d = os.path.join(HERE, "scenes", name)
    sc = O.Scene.from_dir(d)
    lines = open(os.path.join(d, "exportConfig.txt")).read().split("\n")
    lines[1], lines[2], lines[3] = (str(v) for v in GD_LEVELS)
    cfg = O.parse_export_config("\n".join(lines))
    table = O.LookupTable(open(os.path.join(ROOT, "designcsg_amd", "data", "lookupTable.txt")).read())
    c, dia, _ = sc.estimate_bounding_box(cfg.bounding_box_half_diameter)
    tris, _ = sc.get_surface(c, dia[0], cfg, table)
    return sorted_triangles(tris).reshape(-1, 3)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "k2_vectors.npz")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    have = R.build_ref_k2()
    arrays, meta = {}, {"scenes": {}, "generator": "tests/golden/make_k2_vectors.py",
                        "n_uniform": N_UNIFORM, "n_near": N_NEAR, "n_pool": N_POOL,
                        "flags": {"default": "-x cl -cl-std=CL2.0 -DSTACK_MEMORY_PER_PIXEL=64 -w -O3 (gfx950)",
                                  "strict": "default + -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt"}}
    try:   # the GPU agent's block of rocminfo (the first agents are the host CPUs)
        blocks = subprocess.run(["rocminfo"], capture_output=True, text=True).stdout.split("Agent ")
        gpu = [b for b in blocks if "gfx950" in b][0]
        cus = gpu.split("Compute Unit:")[1].split("\n")[0].strip()
        meta["device"] = (gpu.split("Marketing Name:")[1].split("\n")[0].strip() + " (gfx950, %s CUs)" % cus).strip()
    except Exception:
        meta["device"] = "unknown"
    lat = lattice_points()
    for name in VECTOR_SCENES:
        assert name in have, "oracle/_ref/k2_%s_*.hsaco missing: build them in the build container" % name
        d = os.path.join(HERE, "scenes", name)
        t0 = time.time()
        ks = R.RefK2(d, "strict")
        kd = R.RefK2(d, "default")
        pool = pool_points(name)
        s_pool = ks.eval_sdf_at_points(pool, local_size=64)          # selection only: not a stored output
        near = np.sort(np.argpartition(np.abs(s_pool), N_NEAR)[:N_NEAR])   # the N_NEAR pool points closest to the surface
        pts = np.concatenate([uniform_points(name), pool[near]])
        arrays[name + "/near_index"] = near.astype(np.uint32)
        res = {}
        for fl, k in (("strict", ks), ("default", kd)):
            res[fl] = (k.eval_sdf_at_points(pts), k.eval_normal_at_points(pts), k.eval_sdf_at_points(lat))
        # local size 1 (the reference's) and 64 give the same bits: each work-item is a pure function of its point
        assert np.array_equal(ks.eval_sdf_at_points(pts, local_size=64).view(np.uint32), res["strict"][0].view(np.uint32))
        for i, key in enumerate(("sdf", "normal", "lattice_sdf")):
            arrays["%s/strict/%s" % (name, key)] = res["strict"][i]
            small, big, vals = pack_delta(res["strict"][i], res["default"][i])
            arrays["%s/default/%s.d8" % (name, key)] = small
            arrays["%s/default/%s.big_index" % (name, key)] = big
            arrays["%s/default/%s.big_value" % (name, key)] = vals
        m = {"objects": ks.num_objects, "steps": ks.num_steps,
             "hsaco_sha256": {fl: hashlib.sha256(open(R.hsaco_path(name, fl), "rb").read()).hexdigest() for fl in R.FLAVOURS},
             "lattice_negative": {fl: int((res[fl][2] < 0).sum()) for fl in res},
             "lattice_sum": {fl: float(res[fl][2].astype(np.float64).sum()) for fl in res}}
        if name == "design2":
            asis = R.RefK2(d, "asis")     # the reference's committed k2build.cl as it lies there
            assert np.array_equal(asis.eval_sdf_at_points(pts[:20000]).view(np.uint32), res["default"][0][:20000].view(np.uint32))
            asis.close()
            m["k2build_cl_as_committed_equals_default"] = True
            for fl, k in (("strict", ks), ("default", kd)):
                arrays["design2/%s/probe_sdf" % fl] = k.eval_sdf_at_points(np.array(PROBE_SDF, np.float32))
                arrays["design2/%s/probe_normal" % fl] = k.eval_normal_at_points(np.array(PROBE_NORMAL, np.float32))
        if name in STAGE_A_SCENES:
            m["stage_a"] = {}
            for fl, k in (("strict", ks), ("default", kd)):
                c, dia, n = k.estimate_bounding_box(10.0)
                m["stage_a"][fl] = {"center": c, "diameter": dia, "interior": n, "kernel_ms_local1": k.last_kernel_ms}
        if name in GD_SCENES:
            v = raw_mesh_vertices(name)
            m["gd"] = {"levels": list(GD_LEVELS), "steps": GD_STEPS, "vertices": int(len(v)),
                       "input_sha256": hashlib.sha256(v.tobytes()).hexdigest()}
            g = {fl: k.gradient_descent(v, GD_STEPS, local_size=64) for fl, k in (("strict", ks), ("default", kd))}
            arrays[name + "/strict/gd50"] = g["strict"]
            small, big, vals = pack_delta(g["strict"], g["default"])
            arrays[name + "/default/gd50.d8"] = small
            arrays[name + "/default/gd50.big_index"] = big
            arrays[name + "/default/gd50.big_value"] = vals
            for fl in g:
                w = g[fl].astype(np.float64)
                m["gd"][fl] = {"sums": w.sum(0).tolist(), "sum_abs": float(np.linalg.norm(w, axis=1).sum()),
                               "nan": int(np.isnan(w).sum())}
        ks.close()
        kd.close()
        meta["scenes"][name] = m
        print(name, "%.1f s" % (time.time() - t0), json.dumps({k: v for k, v in m.items() if k != "hsaco_sha256"}), flush=True)
    arrays["meta"] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), dtype=np.uint8)
    np.savez_compressed(out_path, **arrays)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main()
