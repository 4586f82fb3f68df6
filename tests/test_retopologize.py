"""Opt-in T-junction repair in the spirit of cms::retopologize (mesh.hpp:432-529).  PARITY UNPINNED by
construction: the reference's function is undefined behaviour (segfaults), and unfinished (its de-duplication helper is
never called) -- oracle_retopologize DEFINES the stage from the function's own ingredients (see oracle/dcsg_oracle.c).
The CPU tests check that definition against its purpose (seams closed, everything else untouched, identity on the
reference's designs as SURVEY a-9 measured for the repaired reference function); the GPU tests hold the HIP kernels to
the oracle bit for bit."""
import numpy as np
import pytest

from conftest import export_config_text, read_scene


def edge_counts(tris):
    """undirected edge (as a pair of vertex bit patterns) -> number of incident triangles"""
    t = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 3, 3)
    v = t.view(np.uint32).reshape(-1, 3, 3)
    cnt = {}
    for tri in v:
        p = [tuple(int(x) for x in q) for q in tri]
        for a, b in ((0, 1), (1, 2), (2, 0)):
            k = (p[a], p[b]) if p[a] <= p[b] else (p[b], p[a])
            cnt[k] = cnt.get(k, 0) + 1
    return cnt


def t_junction_mesh():
    """A big triangle whose edge A-B carries the vertex M of two finer neighbours on the other side: the classic
    T-junction between octree leaves of different levels.  M is the edge midpoint = sample 1 of 2 along the edge."""
    A, B, Cc = np.float32([0, 0, 0]), np.float32([1, 0, 0]), np.float32([0.5, 1, 0])
    M, D = (A + B) / np.float32(2), np.float32([0.5, -1, 0])
    return np.array([[A, B, Cc], [B, M, D], [M, A, D]], dtype=np.float32), (A, B, Cc, M, D)


def test_oracle_retopologize_repairs_a_t_junction_and_keeps_everything_else(oracle):
    tris, (A, B, Cc, M, D) = t_junction_mesh()
    before = edge_counts(tris)
    assert sum(1 for c in before.values() if c == 1) == 7  # 4 outer edges + the three unmatched pieces of the seam
    out = oracle.retopologize(tris, [0.5, 0.0, 0.0], 4.0, 3, 4)  # 2 samples per edge
    # the polygon of the big triangle is (A, M, B, C): strip rule -> (A, M, B) [zero area] and (B, C, A)
    assert len(out) == 4
    assert np.array_equal(out[0], np.array([A, M, B])) and np.array_equal(out[1], np.array([B, Cc, A]))
    assert np.array_equal(out[2:], tris[1:])
    after = edge_counts(out)
    assert sum(1 for c in after.values() if c == 1) == 4  # only the outer boundary is left open: the seam is closed
    # no foreign vertex on any edge -> identity, bit for bit
    lone = tris[:1]
    assert np.array_equal(oracle.retopologize(lone, [0.5, 0.0, 0.0], 4.0, 2, 4).view(np.uint32), lone.view(np.uint32))
    assert len(oracle.retopologize(np.zeros((0, 3, 3), np.float32), [0, 0, 0], 1.0, 2, 4)) == 0


def test_oracle_retopologize_is_identity_on_a_meshed_design(oracle, lookup_table):
    """with its captures repaired the reference's retopologize adds no triangle and moves no vertex on
    Sphere / Design2 at 128^3.  The restatement must show the same."""
    for name in ("sphere", "design2"):
        from conftest import scene_dir
        sc = oracle.Scene.from_dir(scene_dir(name))
        cfg = oracle.parse_export_config(export_config_text(name, levels=(4, 6, 7)))
        c, d, _ = sc.estimate_bounding_box(cfg.bounding_box_half_diameter)
        raw, _ = sc.get_surface(c, d[0], cfg, lookup_table)
        out = oracle.retopologize(raw, c, d[0], 4, 7)
        assert out.shape == raw.shape and np.array_equal(out.view(np.uint32), raw.view(np.uint32))


@pytest.fixture(scope="module")
def ev(built_lib):
    from designcsg_amd.evaluator import Evaluator
    e = Evaluator(0)
    s = read_scene("sphere")
    assert e.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
    yield e
    e.close()


@pytest.mark.gpu
def test_hip_retopologize_equals_the_oracle(ev, oracle):
    rng = np.random.default_rng(12)
    tris, _ = t_junction_mesh()
    # a field of T-junction patches at random places and scales plus random clutter triangles; 8 samples per edge
    patches = []
    for _ in range(3000):
        s, o = np.float32(rng.uniform(0.05, 0.4)), rng.uniform(-3, 3, 3).astype(np.float32)
        patches.append(tris * s + o)
    soup = np.concatenate(patches + [rng.uniform(-4, 4, (5000, 3, 3)).astype(np.float32)]).astype(np.float32)
    for min_level, grid in ((4, 7), (6, 9), (7, 7), (3, 8)):
        want = oracle.retopologize(soup, [0.1, -0.2, 0.05], 9.5, min_level, grid)
        ev.upload_triangles(soup)
        n = ev.retopologize([0.1, -0.2, 0.05], 9.5, min_level, grid)
        got = ev.download_triangles()
        assert n == len(want) and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (min_level, grid)
    with pytest.raises(Exception):
        ev.retopologize([0, 0, 0], 1.0, 1, 9)  # 256 samples per edge: outside what the kernels' keep masks hold


@pytest.mark.gpu
@pytest.mark.parametrize("name,levels", [("sphere", (4, 6, 7)), ("design2", (5, 7, 8)), ("mixed_groups", (3, 6, 7))])
def test_export_with_retopologize(ev, oracle, lookup_table, name, levels):
    """DCSG_EXPORT_RETOPO end to end against the oracle's pipeline (get_surface -> retopologize -> gradient descent),
    and the default export untouched by the option's existence."""
    from designcsg_amd.evaluator import Evaluator
    from conftest import scene_dir
    e = Evaluator(0)
    try:
        s = read_scene(name)
        assert e.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"], s.get("arbitrary_data"))[0] == 0
        cfg_txt = export_config_text(name, levels=levels, gd=5)
        sc = oracle.Scene.from_dir(scene_dir(name))
        cfg = oracle.parse_export_config(cfg_txt)
        c, d, _ = sc.estimate_bounding_box(cfg.bounding_box_half_diameter)
        raw, _ = sc.get_surface(c, d[0], cfg, lookup_table)
        want = sc.gradient_descent(oracle.retopologize(raw, c, d[0], levels[0], levels[2]), 5)
        st = e.export(cfg_txt, retopologize=True)
        got = e.download_triangles()
        key = lambda t: np.sort(np.ascontiguousarray(t, dtype=np.float32).reshape(-1, 9).view(np.uint32).copy().view(
            [("", np.uint32)] * 9), axis=0)
        assert st.triangles == len(want) and np.array_equal(key(got), key(want))
        plain = e.export(cfg_txt)
        assert plain.triangles == len(raw)
        assert np.array_equal(key(e.download_triangles()), key(sc.gradient_descent(raw, 5)))
    finally:
        e.close()
