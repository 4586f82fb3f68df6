"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Bar (floating point): north_star states ".stl diff within 1e-5"; these tests hold the HIP path to
the stricter bar of bit equality with the oracle wherever both pin the arithmetic (IEEE, no
contraction, correctly rounded sqrt/divide), and state 1e-5 where files are compared.
"""
import os

import numpy as np
import pytest

from conftest import ROOT, export_config_text, read_scene, scene_dir

pytestmark = pytest.mark.gpu

ALL_SCENES = ["sphere", "design1", "design2", "mixed_groups", "flat256", "grouped256", "ad_table", "prefabs", "const_table", "global_state", "userbrush128",
              "smooth_blobs", "smooth256"]   # the last two: smooth-join extension, parity vs the oracle only
STL_TOL = 1e-5  # north_star: ".stl diff within 1e-5"


@pytest.fixture(scope="module")
def ev(built_lib):
    from designcsg_amd.evaluator import Evaluator
    e = Evaluator(0)
    yield e
    e.close()


def load(ev, name):
    s = read_scene(name)
    status, log = ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"], s.get("arbitrary_data"))
    assert status == 0, log
    return s


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def sorted_tris(t):
    """Triangle set in a canonical order (the reference's order is thread-interleaved)."""
    b = np.ascontiguousarray(t, dtype=np.float32).reshape(-1, 9).view(np.uint32)
    return b[np.lexsort(b.T[::-1])]


EXPECTED_STACK_DEPTH = {"sphere": 1, "design1": 1, "design2": 1, "flat256": 1, "mixed_groups": 2, "grouped256": 2,
                        "ad_table": 1, "prefabs": 2, "const_table": 1, "global_state": 1, "smooth_blobs": 3, "smooth256": 2, "userbrush128": 1}


# <= 48 object evaluations: inline per-scene code; userbrush128 (128 evaluations of the design's own brushes): the threaded form
SPECIALISED = {"sphere", "design1", "design2", "mixed_groups", "ad_table", "prefabs", "const_table", "global_state", "smooth_blobs", "userbrush128"}


@pytest.mark.parametrize("interpreter", ["auto", "stack", "generic"])
@pytest.mark.parametrize("name", ALL_SCENES)
def test_point_eval_bit_exact(ev, oracle, name, interpreter, monkeypatch):
    """Every device evaluation path -- per-scene specialised code, the register-stack interpreter of the
    stack program, the generic LDS-stack interpreter of the register program -- against the oracle, which
    runs the build procedure exactly as buildprocedure.txt states it."""
    if interpreter == "stack":
        monkeypatch.setenv("DCSG_SPECIALISE", "0")
        monkeypatch.setenv("DCSG_CULL", "0")
    if interpreter == "generic":
        monkeypatch.setenv("DCSG_SPECIALISE", "0")
        monkeypatch.setenv("DCSG_CULL", "0")
        monkeypatch.setenv("DCSG_FORCE_VARIANT", "0")
    load(ev, name)
    info = ev.scene_info()
    assert info["stack_depth"] == EXPECTED_STACK_DEPTH[name]
    # flat256 / grouped256 (built-in brushes, hundreds of objects) get the tile-culling kernels by default
    auto = 3 if name in SPECIALISED else (4 if name in ("flat256", "grouped256", "smooth256") else 1)
    want_variant = {"auto": auto, "stack": 1, "generic": 0}[interpreter]
    assert info["interpreter"] == want_variant
    osc = oracle.Scene.from_dir(scene_dir(name))
    rng = np.random.default_rng(7)
    n = 200_000 if name in ("sphere", "design1", "design2") else 40_000
    pts = rng.uniform(-5.0, 5.0, (n, 3)).astype(np.float32)
    if name.startswith("smooth"):
        pts[n // 2:] *= np.float32(0.5)     # more samples inside the blend zones
    # include the golden probe points of SURVEY 8(c) and exact symmetry points (normalize(0) = 0)
    pts[:6] = [(0, 0, 0), (1, 0, 0), (0, -4.3, 0), (2.5, 2.5, 2.5), (0.3, 0.1, -0.7), (-0.0, 0.0, -0.0)]
    got_s, got_n = ev.eval_sdf_at_points(pts), ev.eval_normal_at_points(pts)
    want_s, want_n = osc.eval_sdf(pts), osc.eval_normal(pts)
    assert np.array_equal(bits(got_s), bits(want_s))
    assert np.array_equal(bits(got_n), bits(want_n))
    # coordinates the fast specialised code must hand to its exact path: non-finite, huge, -0.0 (each mixed with
    # ordinary points in the same wave); NaN results compare as NaN, everything else bit for bit
    odd = rng.uniform(-2.0, 2.0, (256, 3)).astype(np.float32)
    odd[3] = (np.inf, 0.1, 0.2)
    odd[17] = (0.3, -np.inf, 0.0)
    odd[70] = (np.nan, 0.0, 0.0)
    odd[90] = (1e35, 0.0, -1e35)
    odd[131] = (-0.0, 0.25, 0.5)
    odd[200] = (3e38, 3e38, -3e38)
    with np.errstate(all="ignore"):
        for got, want in ((ev.eval_sdf_at_points(odd), osc.eval_sdf(odd)), (ev.eval_normal_at_points(odd), osc.eval_normal(odd))):
            both_nan = np.isnan(got) & np.isnan(want)
            assert np.array_equal(np.where(both_nan, 0, bits(got)), np.where(both_nan, 0, bits(want)))


@pytest.mark.parametrize("interpreter", ["auto", "stack", "generic"])
def test_huge_axis_vectors_keep_the_written_out_arithmetic(ev, oracle, interpreter, monkeypatch, tmp_path):
    """Objects whose reciprocal axis vectors are huge (scales of 1e-4 ... 1e-20): a product of a coefficient and a
    coordinate difference can overflow, and "+ 0 * dx" then is NaN, not nothing -- the specialiser's constant-aware
    simplification (zero terms dropped, T_min onto MAX_DISTANCE as v_min) must not be applied to them.  Every device
    path against the oracle, including coordinates large enough to overflow the transforms."""
    if interpreter != "auto":
        monkeypatch.setenv("DCSG_SPECIALISE", "0")
    if interpreter == "generic":
        monkeypatch.setenv("DCSG_FORCE_VARIANT", "0")
    scene_txt = ("0 1 0.000000 0.000000 0.000000 0.200000 0.000000 0.000000 0.000000 0.200000 0.000000 0.000000 0.000000 0.200000\n"
                 "2 1 0.100000 0.000000 0.000000 30000.000000 0.000000 0.000000 0.000000 0.500000 0.000000 0.000000 0.000000 0.500000\n"
                 "4 1 0.000000 0.300000 0.000000 0.000000 200000000000000000000.000000 0.000000 2.000000 0.000000 0.000000 0.000000 0.000000 100000000000000000000000000000.000000\n"
                 "3 1 -0.200000 0.000000 0.100000 0.700000 0.000000 0.000000 0.000000 0.000000 -4000.000000 0.000000 0.900000 0.000000\n")
    bp = "0 0 0 0\n0 2 1 1\n2 0 1 0\n0 4 2 1\n2 0 1 0\n0 3 3 1\n4 1 -1 1\n3 0 1 0\n1 0 -1 -1"
    assert ev.build(scene_txt, bp, None)[0] == 0
    assert ev.scene_info()["interpreter"] == {"auto": 3, "stack": 1, "generic": 0}[interpreter]
    d = tmp_path / "huge"
    d.mkdir()
    (d / "scene.txt").write_text(scene_txt)
    (d / "buildprocedure.txt").write_text(bp)
    osc = oracle.Scene.from_dir(str(d))
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.uniform(-2, 2, (20000, 3)), rng.uniform(-1e-3, 1e-3, (20000, 3)) + [0.1, 0.3, 0.1],
                          rng.uniform(-1, 1, (2000, 3)) * 1e12, rng.uniform(-1, 1, (2000, 3)) * 1e29]).astype(np.float32)
    pts[:4] = [(0.1, 0.3, 0.0), (0.1, 0.0, 0.0), (-0.2, 0.0, 0.1), (0.0, 0.0, 0.0)]
    with np.errstate(all="ignore"):
        for got, want in ((ev.eval_sdf_at_points(pts), osc.eval_sdf(pts)), (ev.eval_normal_at_points(pts), osc.eval_normal(pts))):
            both_nan = np.isnan(got) & np.isnan(want)
            assert np.array_equal(np.where(both_nan, 0, bits(got)), np.where(both_nan, 0, bits(want)))
        c, dia = [0.0, 0.0, 0.0], 4.0
        gs, gn = ev.eval_lattice(c, dia, 64)
        ws, wn = osc.eval_lattice(c, dia, 64)
        for got, want in ((gs, ws), (gn, wn)):
            both_nan = np.isnan(got) & np.isnan(want)
            assert np.array_equal(np.where(both_nan, 0, bits(got)), np.where(both_nan, 0, bits(want)))


def _camera(yaw, pitch, dist):
    """Orthonormal right / up / forward (float32) and a camera position `dist` behind the origin along forward."""
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    right = np.array([cy, 0.0, -sy], dtype=np.float32)
    up = np.array([sy * sp, cp, cy * sp], dtype=np.float32)
    forward = np.array([sy * cp, -sp, cy * cp], dtype=np.float32)
    return np.array([0.0, 0.0, -dist], dtype=np.float32), right, up, forward


@pytest.mark.parametrize("name", ["sphere", "design1", "design2", "mixed_groups", "ad_table", "const_table", "global_state", "smooth_blobs"])
def test_preview_image_equals_oracle(ev, oracle, name):
    """Kernel k1 (sphere-traced 640x480 preview) against the oracle's restatement, byte for byte:
    default camera of the reference (DrawPane.h:67 with the identity basis) and a rotated, closer one."""
    load(ev, name)
    osc = oracle.Scene.from_dir(scene_dir(name))
    for cam in (_camera(0.0, 0.0, 15.0), _camera(0.6, -0.35, 9.0)):
        got = ev.render_preview(*cam)
        want = osc.render_preview(*cam)
        assert got.shape == want.shape == (480, 640, 3)
        diff = np.argwhere((got != want).any(axis=2))
        assert len(diff) == 0, "first differing pixels (row, col): %s" % diff[:5].tolist()
        assert len(np.unique(got.reshape(-1, 3), axis=0)) > 16     # an image, not a flat colour
    # the scene loaded without scene.cl text: standard brushes and materials
    if name == "sphere":
        s = read_scene(name)
        assert ev.build(s["scene.txt"], s["buildprocedure.txt"], None)[0] == 0
        cam = _camera(0.0, 0.0, 15.0)
        assert np.array_equal(ev.render_preview(*cam), osc.render_preview(*cam))


def test_headless_run_export_and_preview(oracle, tmp_path):
    """python -m designcsg_amd.run design.py --export PREFIX --preview IMAGE: File>Run + File>Export of the GUI,
    headless, in a fresh process (own front-end -> scene files -> libdcsg_hip)."""
    import subprocess
    import sys
    design = os.path.join(ROOT, "tests", "designs", "mixed_groups.py")
    r = subprocess.run([sys.executable, "-m", "designcsg_amd.run", design, "--workdir", str(tmp_path / "wd"),
                        "--export", str(tmp_path / "out"), "--preview", str(tmp_path / "view.ppm")],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "exported" in r.stdout
    osc = oracle.Scene.from_dir(str(tmp_path / "wd"))
    cfg = oracle.parse_export_config((tmp_path / "wd" / "exportConfig.txt").read_text())
    c, d, _ = osc.estimate_bounding_box(cfg.bounding_box_half_diameter)
    table = oracle.LookupTable(open(os.path.join(ROOT, "designcsg_amd", "data", "lookupTable.txt")).read())
    raw, _ = osc.get_surface(c, d[0], cfg, table)
    want = osc.gradient_descent(raw, cfg.gradient_descent_steps)
    stl = (tmp_path / "out.stl").read_bytes()
    assert len(stl) == 84 + 50 * len(want) and os.path.getsize(tmp_path / "out.ply") > 85 * len(want)
    got = np.frombuffer(stl[84:], dtype=np.uint8).reshape(-1, 50)[:, 12:48].copy().view(np.float32).reshape(-1, 3, 3)
    assert np.array_equal(sorted_tris(got[:, :, [0, 2, 1]]), sorted_tris(want))     # the STL stores (x, z, y)
    ppm = (tmp_path / "view.ppm").read_bytes()
    assert ppm.startswith(b"P6\n640 480\n255\n") and len(ppm) == 15 + 640 * 480 * 3
    img = np.frombuffer(ppm[15:], dtype=np.uint8).reshape(480, 640, 3)
    assert np.array_equal(img, osc.render_preview([0, 0, -15], [1, 0, 0], [0, 1, 0], [0, 0, 1]))
    # the same export sharded over "two GPUs" in one process (--devices: dcsg_export_multi; here both contexts on GPU 0)
    r = subprocess.run([sys.executable, "-m", "designcsg_amd.run", design, "--workdir", str(tmp_path / "wd2"),
                        "--export", str(tmp_path / "out2"), "--devices", "0,0"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "on 2 GPUs" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    stl2 = (tmp_path / "out2.stl").read_bytes()
    assert len(stl2) == len(stl) and os.path.getsize(tmp_path / "out2.ply") == os.path.getsize(tmp_path / "out.ply")
    got2 = np.frombuffer(stl2[84:], dtype=np.uint8).reshape(-1, 50)[:, 12:48].copy().view(np.float32).reshape(-1, 3, 3)
    assert np.array_equal(sorted_tris(got2[:, :, [0, 2, 1]]), sorted_tris(want))


@pytest.mark.parametrize("seed", list(range(6)))
def test_random_scenes(ev, oracle, lookup_table, seed, tmp_path, monkeypatch):
    """Differential test on generated designs (tests/designs/random_tree.py: every group kind, smooth joins, carved
    leaves, nesting, axis-aligned objects at the origin): the scene the own front-end compiles is evaluated on every
    device path the scene is eligible for and exported; everything must equal the oracle bit for bit."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "designs"))
    try:
        from random_tree import make_script
    finally:
        sys.path.pop(0)
    from designcsg_amd.run import run_design
    script = tmp_path / "design.py"
    script.write_text(make_script(1000 + seed, big=(seed % 3 == 2)))
    wd = run_design(str(script), str(tmp_path / "wd"))
    osc = oracle.Scene.from_dir(wd)
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-4.5, 4.5, (30_000, 3)).astype(np.float32)
    pts[:3] = [(0, 0, 0), (-0.0, 0.0, 1.0), (2.5, -0.0, -0.0)]
    want_s, want_n = osc.eval_sdf(pts), osc.eval_normal(pts)
    seen = set()
    for env in ({}, {"DCSG_SPECIALISE": "0"}, {"DCSG_SPECIALISE": "0", "DCSG_CULL": "0"},
                {"DCSG_SPECIALISE": "0", "DCSG_CULL": "0", "DCSG_FORCE_VARIANT": "0"}, {"DCSG_NO_SIMPLIFY": "1"}):
        for k in ("DCSG_SPECIALISE", "DCSG_CULL", "DCSG_FORCE_VARIANT", "DCSG_NO_SIMPLIFY"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert ev.build_from_dir(wd)[0] == 0
        seen.add(ev.scene_info()["interpreter"])
        assert np.array_equal(bits(ev.eval_sdf_at_points(pts)), bits(want_s)), env
        assert np.array_equal(bits(ev.eval_normal_at_points(pts)), bits(want_n)), env
    assert 0 in seen and len(seen) >= 2
    for k in ("DCSG_SPECIALISE", "DCSG_CULL", "DCSG_FORCE_VARIANT", "DCSG_NO_SIMPLIFY"):
        monkeypatch.delenv(k, raising=False)
    assert ev.build_from_dir(wd)[0] == 0
    cfg_txt = open(os.path.join(wd, "exportConfig.txt")).read()
    ocfg = oracle.parse_export_config(cfg_txt)
    oc, od, on = osc.estimate_bounding_box(ocfg.bounding_box_half_diameter)
    if not od[0] > 0:
        return          # nothing solid in the probe cube
    gs, gn = ev.eval_lattice(oc, od[0], 24)
    ws, wn = osc.eval_lattice(oc, od[0], 24)
    assert np.array_equal(bits(gs), bits(ws)) and np.array_equal(bits(gn), bits(wn))
    for dense in (False, True):
        st = ev.export(cfg_txt, dense_sweep=dense, dense_normals=dense)
        assert list(st.box_center) == oc and st.box_diameter == od[0] and st.interior_points == on
        raw, ost = osc.get_surface(oc, od[0], ocfg, lookup_table)
        want = osc.gradient_descent(raw, ocfg.gradient_descent_steps)
        assert st.triangles == len(want)
        assert np.array_equal(sorted_tris(ev.download_triangles()), sorted_tris(want))
        for l in range(16):
            assert st.mesh.culled[l] == ost.nodes_culled[l] and st.mesh.subdivided[l] == ost.nodes_subdivided[l]


@pytest.mark.parametrize("seed", [0, 1])
def test_plain_union_culling(ev, oracle, lookup_table, seed, tmp_path):
    """Scenes whose root is a plain union -- single draws and drawn groups (unions and intersections of leaves, joined to
    the root by a union) -- take the culling kernels' order-free rule: an item is dropped when its lower bound over the
    tile lies above the smallest upper bound of ALL items (scene flag bit 1, kernels.hip dcsg_top_min_upper), groups are
    walked for their step masks, the projection keeps its list across steps.  Point evaluation (per-wave lists), lattice
    sweeps (per-brick lists, SDF + normals and SDF only) and the export against the oracle, bit for bit."""
    from designcsg_amd.run import run_design
    rng = np.random.default_rng(77 + seed)
    brushes = ["sphere_brush", "box_brush", "cylinder_brush"]

    def tf(spread, lo, hi):
        p, a, sc = rng.uniform(-spread, spread, 3), rng.uniform(0, np.pi, 3), rng.uniform(lo, hi, 3)
        return "Transform.initial(np.array(%r), %r, %r, %r, np.array(%r))" % (list(np.round(p, 3)), round(a[0], 3), round(a[1], 3),
                                                                                round(a[2], 3), list(np.round(sc, 3)))

    L = ["from DesignCSG import *", "from designlibrary import *", "import numpy as np", ""]
    for i in range(40):
        kind = rng.integers(0, 5)
        if kind <= 2:
            L.append("draw(%s, %s)" % (brushes[i % 3], tf(0.8, 0.05, 0.25)))
        else:
            fn = "drawUnion" if kind == 3 else "drawIntersection"
            kids = ", ".join("Component(%s, %s)" % (brushes[rng.integers(3)], tf(0.15, 0.15, 0.35)) for _ in range(int(rng.integers(2, 6))))
            L.append("%s(%s, transform=%s)" % (fn, kids, tf(0.7, 0.8, 1.2)))
    L += ["setExportConfig(2.0, 3, 5, 6, np.pi / 4, 50, 16, 512, 4096)", "commit()"]
    script = tmp_path / "design.py"
    script.write_text("\n".join(L) + "\n")
    wd = run_design(str(script), str(tmp_path / "wd"))
    assert ev.build_from_dir(wd)[0] == 0
    assert ev.scene_info()["interpreter"] == 4, ev.scene_info()
    assert ev.scene_flags() & 2                       # the root is a plain union
    osc = oracle.Scene.from_dir(wd)
    pts = rng.uniform(-1.5, 1.5, (40_000, 3)).astype(np.float32)
    assert np.array_equal(bits(ev.eval_sdf_at_points(pts)), bits(osc.eval_sdf(pts)))
    assert np.array_equal(bits(ev.eval_normal_at_points(pts)), bits(osc.eval_normal(pts)))
    cfg_txt = open(os.path.join(wd, "exportConfig.txt")).read()
    ocfg = oracle.parse_export_config(cfg_txt)
    oc, od, on = osc.estimate_bounding_box(ocfg.bounding_box_half_diameter)
    assert od[0] > 0
    for r in (24, 75):
        gs, gn = ev.eval_lattice(oc, od[0], r)
        ws, wn = osc.eval_lattice(oc, od[0], r)
        assert np.array_equal(bits(gs), bits(ws)) and np.array_equal(bits(gn), bits(wn))
    gs2, _ = ev.eval_lattice(oc, od[0], 75, normals=False)
    assert np.array_equal(bits(gs2), bits(ws))
    st = ev.export(cfg_txt)
    assert list(st.box_center) == oc and st.box_diameter == od[0] and st.interior_points == on
    raw, _ = osc.get_surface(oc, od[0], ocfg, lookup_table)
    want = osc.gradient_descent(raw, ocfg.gradient_descent_steps)
    assert st.triangles == len(want) and st.triangles > 1000
    assert np.array_equal(sorted_tris(ev.download_triangles()), sorted_tris(want))


def test_builtin_bank_equals_jit_bank(ev, oracle):
    """scene.cl = None selects the HIP-native brushes; the generated scene.cl text must agree with them."""
    s = read_scene("flat256")
    rng = np.random.default_rng(3)
    pts = rng.uniform(-5.0, 5.0, (20_000, 3)).astype(np.float32)
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], None)[0] == 0
    assert not ev.scene_info()["jit"]
    a = ev.eval_sdf_at_points(pts)
    # a preprocessor line forces the compile path without changing any brush
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], "#define DCSG_FORCE_JIT 1\n" + s["scene.cl"])[0] == 0
    assert ev.scene_info()["jit"]
    b = ev.eval_sdf_at_points(pts)
    assert np.array_equal(bits(a), bits(b))


@pytest.mark.parametrize("name", ["sphere", "design2", "mixed_groups", "ad_table", "const_table", "global_state", "smooth_blobs", "flat256", "grouped256", "smooth256",
                                  "userbrush128"])
def test_lattice_sweep_bit_exact(ev, oracle, name):
    load(ev, name)
    osc = oracle.Scene.from_dir(scene_dir(name))
    center, diameter, _ = osc.estimate_bounding_box(10.0)
    res = 48  # not a power of two: exercises the x/w division path
    if name in ("flat256", "grouped256", "smooth256"):  # hundreds of objects: keep the oracle's share short; culling kernels on the GPU
        assert ev.scene_info()["interpreter"] == 4
        assert bool(ev.scene_flags() & 2) == (name == "flat256")   # plain union at the root: the order-free culling rule
        for r in (24, 32, 75):   # 76 points per row: bricks clipped at the far faces, quarters of a brick left empty
            gs, gn = ev.eval_lattice(center, diameter[0], r)
            ws, wn = osc.eval_lattice(center, diameter[0], r)
            assert np.array_equal(bits(gs), bits(ws)) and np.array_equal(bits(gn), bits(wn))
        gs2, _ = ev.eval_lattice(center, diameter[0], 75, normals=False)  # SDF-only form of the same kernel
        assert np.array_equal(bits(gs2), bits(ws))
        return
    for r in (res, 64):
        gs, gn = ev.eval_lattice(center, diameter[0], r)
        ws, wn = osc.eval_lattice(center, diameter[0], r)
        assert np.array_equal(bits(gs), bits(ws))
        assert np.array_equal(bits(gn), bits(wn))
    # a Z slab of the same lattice is the same bits (slab sharding)
    gs2, gn2 = ev.eval_lattice(center, diameter[0], 64, z0=17, z1=40)
    assert np.array_equal(bits(gs2), bits(ws[17:40]))
    assert np.array_equal(bits(gn2), bits(wn[17:40]))
    # SDF-only sweeps (what the export runs when normals are sampled on demand)
    gs3, _ = ev.eval_lattice(center, diameter[0], 64, normals=False)
    assert np.array_equal(bits(gs3), bits(ws))
    gs4, _ = ev.eval_lattice(center, diameter[0], 64, z0=17, z1=40, normals=False)
    assert np.array_equal(bits(gs4), bits(ws[17:40]))
    gs5, _ = ev.eval_lattice(center, diameter[0], 64, z0=3, z1=8, normals=False)
    assert np.array_equal(bits(gs5), bits(ws[3:8]))
    for r in (1, 4, 7, 8):   # tiny lattices
        gs6, _ = ev.eval_lattice(center, diameter[0], r, normals=False)
        assert np.array_equal(bits(gs6), bits(osc.eval_lattice(center, diameter[0], r, normals=False)[0]))


@pytest.mark.parametrize("name", ["sphere", "mixed_groups", "design2", "smooth_blobs"])
@pytest.mark.parametrize("env", [{"DCSG_SPECIALISE": "0", "DCSG_CULL": "0"}, {"DCSG_SPECIALISE": "0", "DCSG_CULL": "0", "DCSG_NO_MULTIPASS": "1"},
                                 {"DCSG_SPECIALISE": "0", "DCSG_CULL": "0", "DCSG_FORCE_VARIANT": "0"}],
                         ids=["stack", "stack_single_pass", "generic"])
def test_interpreter_lattice_sweeps_bit_exact(ev, oracle, name, env, monkeypatch):
    """Dense sweeps through the LDS-table interpreters (what every scene the specialiser refuses runs): rows of at least
    64 points use the axis tables, and the SDF-only sweep of the stack interpreters carries a lane's point of each of a
    run's four passes through ONE interpretation of the program (dcsg_interp<K, 4>).  Whole lattices at 97 and 257
    points per row against the oracle, SDF + normals, SDF only, a Z slab, a box that is not centred."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    load(ev, name)
    assert ev.scene_info()["interpreter"] in ((0,) if "DCSG_FORCE_VARIANT" in env else (1, 2)), ev.scene_info()
    osc = oracle.Scene.from_dir(scene_dir(name))
    center, diameter, _ = osc.estimate_bounding_box(10.0)
    for c, d, res in ((center, diameter[0], 96), ([center[0] + 0.0131, center[1] - 0.0077, center[2] + 0.0253], diameter[0] * 1.037, 256)):
        if res == 256 and name == "design2" and "DCSG_FORCE_VARIANT" in env:
            continue          # the generic interpreter on a 63-box brush: keep the oracle's and the GPU's share short
        ws, wn = osc.eval_lattice(c, d, res)
        gs, gn = ev.eval_lattice(c, d, res)
        assert np.array_equal(bits(gs), bits(ws)) and np.array_equal(bits(gn), bits(wn))
        gs2, _ = ev.eval_lattice(c, d, res, normals=False)
        assert np.array_equal(bits(gs2), bits(ws))
        gs3, _ = ev.eval_lattice(c, d, res, z0=res // 3, z1=res // 3 + 9, normals=False)
        assert np.array_equal(bits(gs3), bits(ws[res // 3:res // 3 + 9]))


@pytest.mark.parametrize("name", ["sphere", "design1", "design2"])
def test_fused_sweep_with_and_without_the_host_neg_zero_check(ev, oracle, name, monkeypatch):
    """The host tells the SDF + normal sweep of per-scene code that no lattice coordinate is -0.0 (DCSG_SWEEP_NO_NEG_ZERO),
    and the kernel then skips its per-pass test for one; with the hint withheld (DCSG_NO_ROWS) it tests every pass.
    Same bits both ways, equal to the oracle's."""
    load(ev, name)
    assert ev.scene_info()["interpreter"] == 3, ev.scene_info()
    osc = oracle.Scene.from_dir(scene_dir(name))
    center, diameter, _ = osc.estimate_bounding_box(10.0)
    c, d, res = [center[0] + 0.0131, center[1] - 0.0077, center[2] + 0.0253], diameter[0] * 1.037, 96
    ws, wn = osc.eval_lattice(c, d, res)
    gs, gn = ev.eval_lattice(c, d, res)
    assert np.array_equal(bits(gs), bits(ws)) and np.array_equal(bits(gn), bits(wn))
    monkeypatch.setenv("DCSG_NO_ROWS", "1")
    hs, hn = ev.eval_lattice(c, d, res)
    assert np.array_equal(bits(hs), bits(ws)) and np.array_equal(bits(hn), bits(wn))


LIGHT_DESIGNS = {
    # standard brushes only, at most four object evaluations (dcsg_spec_small): the sweeps keep the row-invariant part of
    # the seven stencil evaluations across the passes of a row segment
    "sphere_offset": "draw(sphere_brush, Transform.initial(np.array([0.31, -0.17, 0.08]), 0.4, -0.7, 1.1, np.array([0.9, 0.6, 0.75])))",
    "cyl_minus_box": "draw(cylinder_brush, Transform.initial(np.array([0.0, 0.0, 0.0]), 0.0, 0.0, 0.0, np.array([1.0, 1.0, 1.0])))\n"
                     "erase(box_brush, Transform.initial(np.array([0.2, 0.1, -0.3]), 0.7, 0.2, -0.4, np.array([0.5, 0.9, 0.4])))",
    "three_in_group": "drawIntersection(Component(sphere_brush, Transform.initial(np.array([0.1, 0.0, 0.0]), 0.0, 0.0, 0.0, np.array([1.0, 1.0, 1.0]))), "
                      "Component(box_brush, Transform.initial(np.array([0.0, 0.05, 0.0]), 0.3, 0.5, 0.1, np.array([0.8, 0.8, 0.8]))), "
                      "Component(cylinder_brush, Transform.initial(np.array([0.0, 0.0, 0.1]), 1.2, -0.4, 0.6, np.array([0.7, 1.3, 0.7]))), "
                      "transform=Transform.initial(np.array([0.05, -0.02, 0.03]), 0.2, 0.1, -0.3, np.array([1.1, 0.9, 1.0])))",
}


@pytest.mark.parametrize("name", ["sphere"] + sorted(LIGHT_DESIGNS))
def test_light_scene_sweeps_bit_exact(ev, oracle, name, tmp_path):
    """Dense sweeps of per-scene code for light scenes (standard brushes, at most four object evaluations), whose runs
    are cut into row segments that share y and z (dcsg_sweep_body: row_pass).  Whole lattices at 257 points per row
    against the oracle, bit for bit: SDF + normals, SDF only, Z slabs, a box centred on the scene's symmetry planes
    (gradient components exactly zero) and one that is not, the rows' remainder column; then planes of the 513^3
    lattice (two row segments and one remainder point per row)."""
    from designcsg_amd.run import run_design
    if name == "sphere":
        load(ev, name)
        osc = oracle.Scene.from_dir(scene_dir(name))
    else:
        script = tmp_path / "design.py"
        script.write_text("from DesignCSG import *\nimport numpy as np\n" + LIGHT_DESIGNS[name] +
                          "\nsetExportConfig(2.0, 3, 5, 6, np.pi / 4, 6, 16, 512, 4096)\ncommit()\n")
        wd = run_design(str(script), str(tmp_path / "wd"))
        assert ev.build_from_dir(wd)[0] == 0
        osc = oracle.Scene.from_dir(wd)
    info = ev.scene_info()
    assert info["interpreter"] == 3, info      # per-scene specialised code
    center, diameter, _ = osc.estimate_bounding_box(10.0)
    boxes = [(center, diameter[0]), ([center[0] + 0.0131, center[1] - 0.0077, center[2] + 0.0253], diameter[0] * 1.037)]
    for c, d in boxes:
        ws, wn = osc.eval_lattice(c, d, 256)
        gs, gn = ev.eval_lattice(c, d, 256)
        assert np.array_equal(bits(gs), bits(ws))
        assert np.array_equal(bits(gn), bits(wn))
        gs2, _ = ev.eval_lattice(c, d, 256, normals=False)
        assert np.array_equal(bits(gs2), bits(ws))
        gs3, gn3 = ev.eval_lattice(c, d, 256, z0=100, z1=131)
        assert np.array_equal(bits(gs3), bits(ws[100:131])) and np.array_equal(bits(gn3), bits(wn[100:131]))
        gs4, _ = ev.eval_lattice(c, d, 256, z0=255, z1=257, normals=False)
        assert np.array_equal(bits(gs4), bits(ws[255:257]))
    # 513 points per row (two segments + one remainder point): the planes around the middle and the last ones
    for z0, z1 in ((250, 262), (509, 513)):
        ws, wn = osc.eval_lattice(center, diameter[0], 512, z0, z1)
        gs, gn = ev.eval_lattice(center, diameter[0], 512, z0=z0, z1=z1)
        assert np.array_equal(bits(gs), bits(ws)) and np.array_equal(bits(gn), bits(wn))


GOLD_BBOX = {  # centre, diameter, interior points
    "sphere": ([0.0, 0.0, 0.0], 5.3515625, 1376312),
    "design2": ([0.0, -0.0390625, 0.0], 9.3359375, 692566),
    "design1": ([0.0, 0.0, 0.0], 6.6796875, 2868560),
}


@pytest.mark.parametrize("name", ["sphere", "design1", "design2", "mixed_groups", "flat256", "grouped256"])
def test_bounding_box_stage(ev, oracle, name):
    load(ev, name)
    c, d, n = ev.estimate_bounding_box(10.0)
    oc, od, on = oracle.Scene.from_dir(scene_dir(name)).estimate_bounding_box(10.0)
    assert (c, d, n) == (oc, od[0], on)
    if name in GOLD_BBOX:
        assert (c, d, n) == GOLD_BBOX[name]


GOLD_TRIS_128 = {  # levels 4/6/7, per-level triangle counts
    "sphere": {5: 48, 6: 33560},
    "design2": {5: 50, 6: 22866},
    "design1": {4: 48, 6: 33704},
}


@pytest.mark.parametrize("name", ["sphere", "design1", "design2", "mixed_groups", "ad_table", "const_table", "global_state", "smooth_blobs", "userbrush128"])
@pytest.mark.parametrize("dense_normals", [False, True])
def test_mesher_and_projection_bit_exact(ev, oracle, lookup_table, name, dense_normals, tmp_path):
    load(ev, name)
    cfg_txt = export_config_text(name, levels=(4, 6, 7), gd=10)
    stl, ply = tmp_path / "o.stl", tmp_path / "o.ply"
    st = ev.export(cfg_txt, str(stl), str(ply), dense_normals=dense_normals)
    got = ev.download_triangles()

    osc = oracle.Scene.from_dir(scene_dir(name))
    ocfg = oracle.parse_export_config(cfg_txt)
    oc, od, _ = osc.estimate_bounding_box(ocfg.bounding_box_half_diameter)
    raw, ost = osc.get_surface(oc, od[0], ocfg, lookup_table)
    want = osc.gradient_descent(raw, ocfg.gradient_descent_steps)

    assert st.triangles == len(want) == len(got)
    per_level = {l: int(st.mesh.triangles[l]) for l in range(16) if st.mesh.triangles[l]}
    assert per_level == {l: int(ost.triangles[l]) for l in range(32) if ost.triangles[l]}
    if name in GOLD_TRIS_128:
        assert per_level == GOLD_TRIS_128[name]
    for l in range(16):
        assert st.mesh.culled[l] == ost.nodes_culled[l]
        assert st.mesh.subdivided[l] == ost.nodes_subdivided[l]
    # same triangle set, same bits, after the projection too
    assert np.array_equal(sorted_tris(got), sorted_tris(want))

    # files: byte-identical to the oracle's writers for the same triangle order
    oracle.write_stl(str(tmp_path / "w.stl"), got)
    oracle.write_ply(str(tmp_path / "w.ply"), got)
    assert stl.read_bytes() == (tmp_path / "w.stl").read_bytes()
    assert ply.read_bytes() == (tmp_path / "w.ply").read_bytes()
    # and, as north_star words it, the STL vertices agree within 1e-5 (here: exactly)
    v = np.frombuffer(stl.read_bytes()[84:], dtype=np.uint8).reshape(-1, 50)[:, 12:48].copy().view(np.float32)
    w = got.reshape(-1, 3, 3)[:, :, [0, 2, 1]].reshape(-1, 9)  # STL stores (x, z, y)
    assert np.max(np.abs(v - w)) <= STL_TOL


@pytest.mark.parametrize("levels", [(5, 5, 5), (2, 4, 6), (1, 3, 5), (6, 6, 7), (3, 6, 6), (2, 2, 7), (4, 5, 9)])
@pytest.mark.parametrize("name", ["mixed_groups", "design1"])
def test_level_combinations(ev, oracle, lookup_table, name, levels):
    """Octree level triples around the switches of dcsg_export: node-wise sweep at level max(min,2) with nodes of
    2, 3, 5, 9 and 17 points per axis, min below 2, min = max = grid, and (2,2,7) / (4,5,9), where nodes would be 33
    points wide and the dense sweep is used instead.  Mesh and per-level statistics against the oracle."""
    load(ev, name)
    cfg_txt = export_config_text(name, levels=levels, gd=4)
    st = ev.export(cfg_txt)
    got = ev.download_triangles()
    osc = oracle.Scene.from_dir(scene_dir(name))
    ocfg = oracle.parse_export_config(cfg_txt)
    oc, od, _ = osc.estimate_bounding_box(ocfg.bounding_box_half_diameter)
    raw, ost = osc.get_surface(oc, od[0], ocfg, lookup_table)
    want = osc.gradient_descent(raw, 4)
    assert st.triangles == len(want)
    for l in range(16):
        assert st.mesh.culled[l] == ost.nodes_culled[l] and st.mesh.subdivided[l] == ost.nodes_subdivided[l]
        assert st.mesh.triangles[l] == ost.triangles[l]
    assert np.array_equal(sorted_tris(got), sorted_tris(want))
    node_wise = levels[2] - max(levels[0], 2) <= 4
    assert (st.swept_nodes > 0) == node_wise
    if not node_wise:
        assert st.swept_points == st.lattice_points


def test_projection_without_lattice(ev, oracle):
    load(ev, "design2")
    import torch
    rng = np.random.default_rng(11)
    pts = rng.uniform(-4.0, 4.0, (30_000, 3)).astype(np.float32)
    t = torch.from_numpy(pts.copy()).cuda()
    ev.project_vertices_device(t.data_ptr(), len(pts), 7, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = oracle.Scene.from_dir(scene_dir("design2")).gradient_descent(pts.reshape(-1, 3, 3), 7)
    assert np.array_equal(bits(t.cpu().numpy()), bits(want.reshape(-1, 3)))


@pytest.mark.parametrize("levels", [(2, 4, 6), (3, 5, 6), (4, 4, 6), (4, 5, 7), (5, 6, 6), (3, 6, 7), (2, 2, 5), (4, 7, 7), (1, 5, 6)])
def test_node_wise_pattern_over_level_configurations(ev, levels):
    """The node-wise sweep writes only the points of a node that the mesher can read, a rule in the three levels of the export
    configuration (kernels.hip, dcsg_in_sweep_pattern): corners and centres down to the maximum level (steps of 1, 2, 4 between
    them), edge-test samples only when some level makes that test (maximum > sweep level), nodes of 3^3 ... 17^3 points.  For
    every shape of the rule the mesh and the level statistics equal those of a dense sweep, where the rule does not exist."""
    for name in ("design1", "flat256", "smooth256"):
        load(ev, name)
        cfg_txt = export_config_text(name, levels=levels, gd=2)
        a = ev.export(cfg_txt)
        ta = sorted_tris(ev.download_triangles())
        b = ev.export(cfg_txt, dense_sweep=True)
        tb = sorted_tris(ev.download_triangles())
        assert a.triangles == b.triangles and np.array_equal(ta, tb), (name, levels)
        for f in ("culled", "subdivided", "leaves", "triangles"):
            assert list(getattr(a.mesh, f)) == list(getattr(b.mesh, f)), (name, levels, f)
        if levels[2] - max(levels[0], 2) <= 4:
            assert a.swept_nodes > 0 and b.swept_nodes == 0


@pytest.mark.parametrize("name", ["design2", "flat256", "smooth256"])
def test_export_switches_change_nothing(ev, name, monkeypatch, tmp_path):
    """The round-4 restructurings of the export keep a switch each for measurements.  Every one of them must leave the mesh,
    the level statistics, stage A and the files exactly as they are: the first levels of the mesher with a host round trip
    each (DCSG_MESH_SYNC_LEVELS), every sample asking the node mask (DCSG_MESH_MASK_ALWAYS), on-demand samples through the
    whole program instead of the cell lists (DCSG_MESH_NO_CELLS), a list per node instead of per eight siblings
    (DCSG_NODE_GROUP_1), a block per node (DCSG_NODE_BLOCKS_PER_CU=0), every point of a swept node instead of those the mesher
    reads (DCSG_SWEEP_ALL_POINTS), a vertex per lane in the projection (DCSG_PROJECT_VERTEX_PER_LANE), other grids of stage A (its result is kept in
    16 copies), normals swept densely, no export pipeline."""
    cfg_txt = export_config_text(name, levels=(4, 6, 7), gd=3)

    def run(env, dense=False):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        load(ev, name)
        stl, ply = tmp_path / "a.stl", tmp_path / "a.ply"
        st = ev.export(cfg_txt, str(stl), str(ply), dense_normals=dense)
        out = (sorted_tris(ev.download_triangles()), [int(v) for v in st.mesh.nodes] + [int(v) for v in st.mesh.culled] +
               [int(v) for v in st.mesh.leaves] + [int(v) for v in st.mesh.triangles], tuple(st.box_center), st.box_diameter, st.interior_points,
               stl.read_bytes(), ply.read_bytes())
        for k in env:
            monkeypatch.delenv(k)
        return out

    want = run({})
    assert len(want[0]) > 1000
    for env in ({"DCSG_MESH_SYNC_LEVELS": "1"}, {"DCSG_MESH_MASK_ALWAYS": "1"}, {"DCSG_MESH_NO_CELLS": "1"}, {"DCSG_NODE_GROUP_1": "1"},
                {"DCSG_NODE_BLOCKS_PER_CU": "0"}, {"DCSG_SWEEP_ALL_POINTS": "1"}, {"DCSG_PROJECT_VERTEX_PER_LANE": "1"}, {"DCSG_BBOX_BLOCKS_PER_CU": "3"}, {"DCSG_BBOX_BLOCKS_PER_CU": "32"}, {"DCSG_NO_PIPELINE": "1"}):
        got = run(env)
        assert np.array_equal(got[0], want[0]), env
        assert got[1:5] == want[1:5], env
        assert got[5] == want[5] and got[6] == want[6], env
    got = run({}, dense=True)
    assert np.array_equal(got[0], want[0]) and got[1:] == want[1:]


@pytest.mark.parametrize("name", ["flat256", "grouped256", "smooth256", "mixed_groups"])
def test_cell_list_projection(ev, oracle, name, monkeypatch):
    """Many-object scenes project through the cell lists (kernels.hip "Cell lists": a list of steps per cell of a cube
    over the scene, each LANE interpreting the list of the cell its vertex is in, idle lanes refilled from a queue).
    Against the oracle, bit for bit, on vertices in no order at all: uniform in the scene, far outside the cube (the
    pseudo-cell of the whole program), on cell faces, with -0.0 coordinates, NaN and infinity; few and many steps, several
    grid resolutions and refill thresholds."""
    import torch
    monkeypatch.setenv("DCSG_CULL", "8")      # mixed_groups (21 objects) takes the culling kernels too ...
    monkeypatch.setenv("DCSG_SPECIALISE", "0")  # ... when it is not given per-scene code
    rng = np.random.default_rng(5)
    pts = rng.uniform(-4.5, 4.5, (20_000, 3)).astype(np.float32)
    pts[:2000] = rng.uniform(-40, 40, (2000, 3))                  # mostly outside the cube
    pts[2000:2600] = np.round(pts[2000:2600] * 8) / 8              # on faces of cells of many grids
    pts[2600:2700, 0] = -0.0
    pts[2700:2800, 1] = -0.0
    pts[2800:2900] = (-0.0, -0.0, -0.0)
    pts[2900] = (np.nan, 0.1, 0.2)
    pts[2901] = (np.inf, 0.1, 0.2)
    pts[2902] = (0.3, -np.inf, 1e30)
    pts[2903] = (1e20, -1e20, 3e38)
    osc = oracle.Scene.from_dir(scene_dir(name))
    stream = torch.cuda.current_stream().cuda_stream

    def same(a, b):
        both_nan = np.isnan(a) & np.isnan(b)
        return np.array_equal(np.where(both_nan, 0, bits(a)), np.where(both_nan, 0, bits(b)))

    with np.errstate(all="ignore"):
        for steps, n in ((1, 20_000), (9, 20_000), (50, 6_000)):
            want = osc.gradient_descent(pts[:n - n % 3].reshape(-1, 3, 3), steps).reshape(-1, 3)
            # (DCSG_CELLS_CAPACITY: the entry buffer starts far too small and has to grow while cells are being built)
            for env in ({}, {"DCSG_CELLS": "16"}, {"DCSG_CELLS": "97", "DCSG_PROJECT_REFILL": "1"}, {"DCSG_PROJECT_REFILL": "64"},
                        {"DCSG_CELLS": "32", "DCSG_CELLS_CAPACITY": "3000"}):
                if steps == 50 and env:
                    continue
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                load(ev, name)      # the cell lists belong to the loaded scene: a new load builds them again
                assert ev.scene_info()["interpreter"] == 4
                t = torch.from_numpy(pts[:len(want)].copy()).cuda()
                ev.project_vertices_device(t.data_ptr(), len(want), steps, stream)
                torch.cuda.synchronize()
                assert same(t.cpu().numpy(), want), (name, steps, env)
                for k in env:
                    monkeypatch.delenv(k)
        # cells are built where vertices go and live as long as the scene: one region, then another (new cells are appended
        # behind the first ones), then the first again -- with an export in between, which reuses the context's small buffers
        load(ev, name)
        halves = [pts[3000:9000][pts[3000:9000, 0] < 0][:1500], pts[3000:9000][pts[3000:9000, 0] >= 0][:1500]]
        wants = [osc.gradient_descent(h.reshape(-1, 3, 3), 6).reshape(-1, 3) for h in halves]
        for which in (0, 1, 0):
            t = torch.from_numpy(halves[which].copy()).cuda()
            ev.project_vertices_device(t.data_ptr(), len(halves[which]), 6, stream)
            torch.cuda.synchronize()
            assert same(t.cpu().numpy(), wants[which]), (name, "region", which)
            if which == 0:
                built = ev.debug_cells()
                assert built["grid"] == 128 and built["cells_built"] > 0
                ev.export(export_config_text(name, levels=(3, 5, 6), gd=2))
            else:
                assert ev.debug_cells()["cells_built"] > built["cells_built"]


@pytest.mark.parametrize("n_slabs", [2, 4, 8])
def test_z_slab_meshing_equals_whole(ev, oracle, n_slabs):
    """Each virtual rank holds only its own planes (+1 halo) and meshes its slab; the union is the
    single-device mesh; out-of-slab samples of coarse nodes are evaluated on demand."""
    import torch
    from designcsg_amd.evaluator import parse_export_config
    load(ev, "design2")
    cfg = parse_export_config(export_config_text("design2", levels=(4, 6, 7)))
    c, d, _ = ev.estimate_bounding_box(cfg.bounding_box_half_diameter)
    res = 1 << cfg.grid_level
    n1 = res + 1
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.empty(n1 * n1 * n1, dtype=torch.float32, device="cuda")
    ev.sweep_lattice_device(c, d, res, 0, n1, full.data_ptr(), 0, stream)
    n_whole, st_whole = ev.mesh_lattice_device(c, d, cfg, full.data_ptr(), 0, 0, res, 0, res, stream)
    whole = ev.download_triangles()
    parts, tri_sum = [], np.zeros(16, dtype=np.int64)
    for r in range(n_slabs):
        lo, hi = r * res // n_slabs, (r + 1) * res // n_slabs
        slab = torch.empty((hi - lo + 1) * n1 * n1, dtype=torch.float32, device="cuda")
        ev.sweep_lattice_device(c, d, res, lo, hi + 1, slab.data_ptr(), 0, stream)
        n, st = ev.mesh_lattice_device(c, d, cfg, slab.data_ptr(), 0, lo, hi, lo, hi, stream)
        parts.append(ev.download_triangles())
        tri_sum += np.array(st.triangles[:16], dtype=np.int64)
        for l in range(16):
            assert st.culled[l] <= st_whole.culled[l]
    assert sum(len(p) for p in parts) == n_whole
    assert np.array_equal(tri_sum, np.array(st_whole.triangles[:16], dtype=np.int64))
    assert np.array_equal(sorted_tris(np.concatenate(parts)), sorted_tris(whole))


def test_error_behaviour(built_lib):
    from designcsg_amd import _lib
    from designcsg_amd.evaluator import Evaluator
    e = Evaluator(0)
    try:
        with pytest.raises(_lib.DcsgError) as ei:  # "Cannot Export Yet.", DesignCSG.cpp:786
            e.eval_sdf_at_points(np.zeros((1, 3), np.float32))
        assert ei.value.code == _lib.ERR_NO_SCENE
        s = read_scene("design2")
        status, log = e.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"].replace("hilbert_sdf(v)", "no_such_fn(v)"))
        assert status == -1 and "no_such_fn" in log  # Evaluator::build -> (-1, build log)
        with pytest.raises(_lib.DcsgError) as ei:  # object index out of range must not reach a kernel
            e.build(s["scene.txt"], "0 0 0 0\n0 2 99 1\n2 0 1 0\n1 0 -1 -1", None)
        assert ei.value.code == _lib.ERR_INVALID
        with pytest.raises(_lib.DcsgError):  # brush 5 without scene.cl
            e.build(s["scene.txt"], s["buildprocedure.txt"], None)
        # empty and ragged inputs
        assert e.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
        assert e.eval_sdf_at_points(np.zeros((0, 3), np.float32)).shape == (0,)
        assert e.eval_normal_at_points(np.zeros((0, 3), np.float32)).shape == (0, 3)
        assert e.eval_sdf_at_points(np.zeros((257, 3), np.float32)).shape == (257,)
    finally:
        e.close()


def test_jit_of_a_new_brush(ev, oracle, tmp_path):
    """A brush nobody pre-compiled (comgr JIT on the GPU box), with an unpinned built-in (sin)."""
    s = read_scene("sphere")
    new_brush = """
        float sd5( float3 v){
            float r = length((float2)(v.x,v.z)) - 0.35;
            return length((float2)(r, v.y)) - 0.12 + 0.01*sin(20.0f*v.x);
        }
    """
    cl = s["scene.cl"].replace("float sdf_bank(", new_brush + "\n        float sdf_bank(") \
                      .replace("switch(shape_id){", "switch(shape_id){\ncase 5: return sd5(v); break;\n")
    scene_txt = s["scene.txt"].replace("\n2 1 ", "\n5 1 ")
    bp = s["buildprocedure.txt"].replace("0 2 1 1", "0 5 1 1")
    assert ev.build(scene_txt, bp, cl)[0] == 0 and ev.scene_info()["jit"]
    p = tmp_path / "scene.cl"
    p.write_text(cl)
    osc = oracle.Scene(scene_txt, bp, str(p))
    rng = np.random.default_rng(5)
    pts = rng.uniform(-3.0, 3.0, (50_000, 3)).astype(np.float32)
    got, want = ev.eval_sdf_at_points(pts), osc.eval_sdf(pts)
    # sin() comes from the ROCm device library on the GPU and libm on the host: unpinned, 1e-6 abs
    assert np.max(np.abs(got - want)) <= 1e-6


@pytest.mark.parametrize("module", ["builtin", "scene"])
def test_sqrt_all_float_inputs(ev, module):
    """The kernels' square root (v_rsq_f32 + one FMA residual correction for waves whose inputs are all ordinary and
    >= 2^-96, the compiler's IEEE sequence otherwise) against IEEE sqrtf for EVERY binary32 bit pattern (builtin module) /
    every 8th block of 2^26 (scene module, where the brush prelude's copy is compared with the kernels' as well)."""
    import torch
    if module == "scene":
        load(ev, "design2")
    else:
        s = read_scene("flat256")
        assert ev.build(s["scene.txt"], s["buildprocedure.txt"], None)[0] == 0 and not ev.scene_info()["jit"]
    stream = torch.cuda.current_stream().cuda_stream
    chunk = 1 << 26
    step = 1 if module == "builtin" else 8
    out = torch.empty(chunk, dtype=torch.float32, device="cuda")
    for c in range(0, 1 << 32, chunk * step):
        bits = torch.arange(c, c + chunk, dtype=torch.int64, device="cuda").to(torch.uint32 if hasattr(torch, "uint32") else torch.int64)
        x = (torch.arange(c, c + chunk, dtype=torch.int64, device="cuda") - (1 << 32) * (c >= (1 << 31))).to(torch.int32).view(torch.float32)
        ev.debug_sqrt_device(x.data_ptr(), chunk, out.data_ptr(), stream)
        torch.cuda.synchronize()
        xin = x.cpu().numpy()
        with np.errstate(invalid="ignore"):
            want = np.sqrt(xin)
        got = out.cpu().numpy()
        nan = np.isnan(want)
        assert np.array_equal(np.isnan(got), nan), hex(c)
        assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32)), hex(c)
        del bits


def _normal_tail_numpy(d, dy, dz):
    """get_normal's tail (k2.cl:175-177 + normalize pinned as v / length(v)) in IEEE float / double arithmetic."""
    k = 1.0 / np.float64(np.float32(2.0 * 0.005))
    with np.errstate(all="ignore"):
        n = [(k * c.astype(np.float64)).astype(np.float32) for c in (d, dy, dz)]
        q = n[0] * n[0]
        q = q + n[1] * n[1]
        q = q + n[2] * n[2]
        ln = np.sqrt(q)
        zero = ln == 0
        return [np.where(zero, np.float32(0), c / ln) for c in n]


def _normal_tail_torch(d, dy, dz):
    """The same on the GPU with every rounding made explicit: products / sums as separate float kernels, square root and
    quotients in double and narrowed (innocuous double rounding for normal results: 53 >= 2 * 24 + 2)."""
    import torch
    k = 1.0 / float(np.float64(np.float32(2.0 * 0.005)))
    n = [torch.mul(c.double(), k).float() for c in (d, dy, dz)]
    q = torch.mul(n[0], n[0])
    q = torch.add(q, torch.mul(n[1], n[1]))
    q = torch.add(q, torch.mul(n[2], n[2]))
    ln = torch.sqrt(q.double()).float()
    zero = ln == 0
    return [torch.where(zero, torch.zeros_like(c), (c.double() / ln.double()).float()) for c in n]


@pytest.mark.parametrize("module", ["builtin", "scene"])
def test_normal_all_float_inputs(ev, module):
    """The tail of get_normal as the kernels compute it -- gradient scale by float-only arithmetic where that is exact,
    in double where a wave holds a component below 2^-60 after scaling -- for EVERY binary32 bit pattern of the first
    gradient component (builtin module; every 8th block of 2^26 for the scene module), alone in its waves and next
    to ordinary lanes.  Reference: numpy float / double arithmetic for the blocks with tiny |D| (subnormal quotients),
    explicit double-rounded torch arithmetic on the GPU elsewhere, cross-checked against numpy on one block."""
    import torch
    if module == "scene":
        load(ev, "design2")
    else:
        s = read_scene("flat256")
        assert ev.build(s["scene.txt"], s["buildprocedure.txt"], None)[0] == 0 and not ev.scene_info()["jit"]
    stream = torch.cuda.current_stream().cuda_stream
    chunk = 1 << 26
    step = 1 if module == "builtin" else 8
    out = torch.empty(3 * chunk, dtype=torch.float32, device="cuda")
    dy, dz = 0.25, -0.5
    odd = (torch.arange(chunk, device="cuda") & 1).bool()
    for ci, c in enumerate(range(0, 1 << 32, chunk * step)):
        x = (torch.arange(c, c + chunk, dtype=torch.int64, device="cuda") - (1 << 32) * (c >= (1 << 31))).to(torch.int32).view(torch.float32)
        for mixed in ((0, 1) if ci % 4 == 0 else (0,)):
            ev.debug_normal_device(x.data_ptr(), chunk, dy, dz, mixed, out.data_ptr(), stream)
            torch.cuda.synchronize()
            d = torch.where(odd, torch.full_like(x, 1.0), x) if mixed else x
            y = torch.where(odd, torch.full_like(x, 2.0), torch.full_like(x, dy)) if mixed else torch.full_like(x, dy)
            z = torch.where(odd, torch.full_like(x, 3.0), torch.full_like(x, dz)) if mixed else torch.full_like(x, dz)
            got = out.view(chunk, 3)
            tiny = (c & 0x7fffffff) < (1 << 28)          # |D| < 2^-95: quotients can be subnormal
            if tiny or c == 0x40000000:
                want = _normal_tail_numpy(d.cpu().numpy(), y.cpu().numpy(), z.cpu().numpy())
                g = got.cpu().numpy()
                for k in range(3):
                    nan = np.isnan(want[k])
                    assert np.array_equal(np.isnan(g[:, k]), nan), (hex(c), mixed, k)
                    assert np.array_equal(g[:, k][~nan].view(np.uint32), want[k][~nan].view(np.uint32)), (hex(c), mixed, k)
            if not tiny:
                want = _normal_tail_torch(d, y, z)
                for k in range(3):
                    w, g = want[k], got[:, k]
                    same = (w.view(torch.int32) == g.view(torch.int32)) | (torch.isnan(w) & torch.isnan(g))
                    assert bool(same.all()), (hex(c), mixed, k, int((~same).sum()))


@pytest.mark.parametrize("module", ["builtin", "scene"])
def test_normal_zero_components(ev, module):
    """Gradients with exactly zero components (lattice points on a symmetry plane of the scene) stay on normalize's
    fast path: +0 and -0 in every position, next to ordinary lanes, against explicitly rounded double arithmetic, sign of
    zero included."""
    import torch
    if module == "scene":
        load(ev, "design2")
    else:
        s = read_scene("flat256")
        assert ev.build(s["scene.txt"], s["buildprocedure.txt"], None)[0] == 0 and not ev.scene_info()["jit"]
    stream = torch.cuda.current_stream().cuda_stream
    n = 1 << 22
    out = torch.empty(3 * n, dtype=torch.float32, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.rand(n, generator=g, device="cuda") + 0.5) * torch.exp2(torch.randint(-12, 13, (n,), generator=g, device="cuda").float())
    x = x * (torch.randint(0, 2, (n,), generator=g, device="cuda").float() * 2 - 1)
    i = torch.arange(n, device="cuda")
    x = torch.where(i % 4 == 0, torch.zeros_like(x), x)
    x = torch.where(i % 4 == 1, -torch.zeros_like(x), x)
    assert int((x.view(torch.int32) == -(1 << 31)).sum()) == n // 4     # the -0.0 lanes are really -0.0
    for dy, dz in ((0.25, -0.5), (0.0, 3.0), (-0.0, 3.0), (2.0, 0.0), (0.0, -0.0), (0.0, 0.0), (-0.0, -0.0)):
        ev.debug_normal_device(x.data_ptr(), n, dy, dz, 0, out.data_ptr(), stream)
        torch.cuda.synchronize()
        want = _normal_tail_torch(x, torch.full_like(x, dy), torch.full_like(x, dz))
        got = out.view(n, 3)
        for k in range(3):
            w, q = want[k], got[:, k]
            same = (w.view(torch.int32) == q.view(torch.int32)) | (torch.isnan(w) & torch.isnan(q))
            assert bool(same.all()), (dy, dz, k, int((~same).sum()))

def test_normal_random_gradients(ev):
    """normalize's quotients take one residual correction instead of the IEEE sequence's two (DESIGN.md section 3,
    tools/div_short_probe.hip has the exhaustive mantissa-pair run): 2^30 random gradients of ordinary magnitude through the
    product kernel against explicitly rounded double arithmetic."""
    import torch
    s = read_scene("flat256")
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], None)[0] == 0
    stream = torch.cuda.current_stream().cuda_stream
    n = 1 << 26
    out = torch.empty(3 * n, dtype=torch.float32, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(11)
    rng = np.random.default_rng(12)
    for it in range(16):
        # mantissas uniform, exponents spread over 2^-20 .. 2^20
        x = (torch.rand(n, generator=g, device="cuda") + 1.0) * torch.exp2(torch.randint(-20, 21, (n,), generator=g, device="cuda").float())
        x = x * (torch.randint(0, 2, (n,), generator=g, device="cuda").float() * 2 - 1)
        dy, dz = [float(np.float32(rng.uniform(1, 2) * 2.0 ** rng.integers(-20, 21) * rng.choice([-1, 1]))) for _ in range(2)]
        ev.debug_normal_device(x.data_ptr(), n, dy, dz, 0, out.data_ptr(), stream)
        torch.cuda.synchronize()
        want = _normal_tail_torch(x, torch.full_like(x, dy), torch.full_like(x, dz))
        got = out.view(n, 3)
        for k in range(3):
            same = want[k].view(torch.int32) == got[:, k].view(torch.int32)
            assert bool(same.all()), (it, k, int((~same).sum()))


def test_cold_build_exports_before_the_rest_part_is_ready(built_lib, oracle, tmp_path, monkeypatch):
    """A scene module is two code objects (csrc/jit.hpp): a cold dcsg_load_scene compiles the export part, returns, and a
    background host thread compiles the rest (dense sweeps, point evaluation) while the caller already exports.  The
    export's files must be byte-identical to those of a warm load, and the rest part -- joined and loaded on first use --
    must evaluate like the oracle.  DCSG_JIT_SYNC=1 (everything in the foreground) gives the same."""
    from designcsg_amd.evaluator import Evaluator
    s = read_scene("design1")
    cfg_txt = export_config_text("design1", levels=(3, 5, 6), gd=6)
    osc = oracle.Scene.from_dir(scene_dir("design1"))
    pts = np.random.default_rng(3).uniform(-3, 3, (5000, 3)).astype(np.float32)
    want = osc.eval_sdf(pts)
    files = {}
    for mode in ("warm", "cold", "cold_sync"):
        monkeypatch.delenv("DCSG_JIT_NO_CACHE", raising=False)
        monkeypatch.delenv("DCSG_JIT_SYNC", raising=False)
        if mode != "warm":
            monkeypatch.setenv("DCSG_JIT_NO_CACHE", "1")
        if mode == "cold_sync":
            monkeypatch.setenv("DCSG_JIT_SYNC", "1")
        e = Evaluator(0)
        try:
            assert e.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
            prof = e.load_profile()["load_scene"]
            assert ("code_object_compiled" in prof) == (mode != "warm"), prof
            assert ("rest_part_compiled_and_loaded" in prof) == (mode == "cold_sync"), prof
            stl, ply = tmp_path / (mode + ".stl"), tmp_path / (mode + ".ply")
            st = e.export(cfg_txt, str(stl), str(ply))       # needs the export part only
            assert st.triangles > 0
            files[mode] = (stl.read_bytes(), ply.read_bytes())
            got = e.eval_sdf_at_points(pts)                   # needs the rest part: joins the background build
            assert np.array_equal(bits(got), bits(want)), mode
            gs, gn = e.eval_lattice([0.0, 0.0, 0.0], 6.0, 24)
            ws, wn = osc.eval_lattice([0.0, 0.0, 0.0], [6.0] * 3, 24)
            assert np.array_equal(bits(gs), bits(ws)) and np.array_equal(bits(gn), bits(wn)), mode
        finally:
            e.close()
    assert files["cold"] == files["warm"] == files["cold_sync"]


def test_set_arbitrary_data_after_build(ev, oracle):
    """Evaluator::setArbitraryData (Evaluator.cpp:213-225) on a loaded scene: the brushes see the new table."""
    s = load(ev, "ad_table")
    rng = np.random.default_rng(8)
    pts = rng.uniform(-3.0, 3.0, (20_000, 3)).astype(np.float32)
    a = ev.eval_sdf_at_points(pts)
    table = np.zeros(131072, dtype=np.float32)
    table[:len(s["arbitrary_data"])] = s["arbitrary_data"]
    table[36:48] *= np.float32(1.5)  # AD_BLOB_RADII
    ev.setArbitraryData(table)
    b = ev.eval_sdf_at_points(pts)
    osc = oracle.Scene(s["scene.txt"], s["buildprocedure.txt"], scene_dir("ad_table") + "/scene.cl", table)
    assert not np.array_equal(a, b)
    assert np.array_equal(bits(b), bits(osc.eval_sdf(pts)))


def test_large_object_tables(built_lib, oracle, lookup_table, monkeypatch):
    """Beyond the reference's MAX_OBJECTS 512 / MAX_BUILD_STEPS 256 (DrawPane.h:14-15): the tables live in LDS, so
    ~1400 objects fit (160 KiB); more is refused with a clear error instead of overflowing.  The 1200-object scene also
    EXPORTS (the mesher's kernels take no LDS beyond the scene's own area), the mesh checked against the oracle with the
    on-demand samples through the cell lists, through the whole program, and with normals read from a dense sweep."""
    from designcsg_amd import _lib
    from designcsg_amd.evaluator import Evaluator
    rng = np.random.default_rng(0)

    def scene(n):
        rows, bp = ["0 1 0 0 0 0.2 0 0 0 0.2 0 0 0 0.2"], ["0 0 0 0"]
        for i in range(n):
            p, s = rng.uniform(-4, 4, 3), 1.0 / rng.uniform(0.2, 0.6)
            rows.append("2 1 %.6f %.6f %.6f %.6f 0 0 0 %.6f 0 0 0 %.6f" % (p[0], p[1], p[2], s, s, s))
            bp += ["0 2 %d 1" % (i + 1), "2 0 1 0"]
        return "\n".join(rows) + "\n", "\n".join(bp + ["1 0 -1 -1"])

    e = Evaluator(0)
    try:
        st, bp = scene(1200)
        assert e.build(st, bp, None)[0] == 0
        pts = rng.uniform(-5, 5, (4000, 3)).astype(np.float32)
        osc = oracle.Scene(st, bp)
        assert np.array_equal(bits(e.eval_sdf_at_points(pts)), bits(osc.eval_sdf(pts)))
        gs, gn = e.eval_lattice([0.0, 0.0, 0.0], 10.0, 16)
        ws, wn = osc.eval_lattice([0.0, 0.0, 0.0], 10.0, 16)
        assert np.array_equal(bits(gs), bits(ws)) and np.array_equal(bits(gn), bits(wn))
        cfg_txt = "10.0\n2\n4\n5\n0.7853981633974483\n3\n16\n512\n4096\n"
        stats = e.export(cfg_txt)
        cfg = oracle.parse_export_config(cfg_txt)
        c, d, _ = osc.estimate_bounding_box(cfg.bounding_box_half_diameter)
        raw, _ = osc.get_surface(c, d[0], cfg, lookup_table)
        want = osc.gradient_descent(raw, 3)
        assert stats.triangles == len(want) > 1000
        assert np.array_equal(sorted_tris(e.download_triangles()), sorted_tris(want))
        # the mesher's on-demand samples of such a scene read the cell lists; the same mesh without them (every sample runs the
        # whole program) and with densely swept normals (the complex-edge test then reads the lattice)
        for env, dense in (("DCSG_MESH_NO_CELLS", False), (None, True)):
            if env:
                monkeypatch.setenv(env, "1")
            assert e.export(cfg_txt, dense_normals=dense).triangles == len(want)
            assert np.array_equal(sorted_tris(e.download_triangles()), sorted_tris(want))
            if env:
                monkeypatch.delenv(env)
        st, bp = scene(2400)
        with pytest.raises(_lib.DcsgError) as ei:
            e.build(st, bp, None)
        assert ei.value.code == _lib.ERR_INVALID and "LDS" in ei.value.message
    finally:
        e.close()
