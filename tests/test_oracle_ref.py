"""The oracle and the product's host code against the REAL reference, where the reference builds on its own.

oracle/_ref/libdcsg_ref.so (oracle/Makefile target `ref`, built by __graft_entry__.build() wherever /root/reference
exists; the built file travels to the GPU box) wraps the reference's stand-alone headers -- utils.hpp (both writers),
readLookupTable.hpp, geometry.hpp, octree.hpp -- and its lattice cache ISV::ISV3D64 (ISV.hpp, instantiated as the export
driver does; oracle/ref_isv_shim.cpp) without any stand-in header or library.  Its mesher and Evaluator cannot be built
here (wx / Win32 headers, OpenCL runtime), so those parts of the oracle stay pinned by the survey's golden vectors
(tests/test_oracle_golden.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, TABLE

REF = os.path.join(ROOT, "oracle", "_ref", "libdcsg_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (no /root/reference here)")


@pytest.fixture(scope="module")
def ref():
    L = C.CDLL(REF)
    L.ref_write_stl.argtypes = [C.c_char_p, C.c_void_p, C.c_ulonglong]
    L.ref_write_ply.argtypes = [C.c_char_p, C.c_void_p, C.c_ulonglong]
    L.ref_lookup_table.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
    L.ref_box_corners.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    L.ref_subdivide.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ref_angle_between.restype = C.c_float
    L.ref_angle_between.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
    L.ref_magnitude.restype = C.c_float
    L.ref_magnitude.argtypes = [C.c_void_p]
    L.ref_isv_coords.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_ulonglong, C.c_void_p]
    L.ref_isv_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_ulonglong, C.c_void_p]
    L.ref_isv_values.restype = C.c_longlong
    L.ref_isv_values.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_ulonglong,
                                 ISV_EVAL, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def p(a):
    return a.ctypes.data_as(C.c_void_p)


ISV_EVAL = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_ulonglong, C.POINTER(C.c_float), C.c_void_p)

# lattice boxes: the reference's own stage-A results for Sphere and Design2 ("nice": multiples of 10/256), a box that is
# nothing of the sort, and a non-cubic one (ISV3D64 takes three diameters; the export always passes a cube)
ISV_BOXES = [([0.0, 0.0, 0.0], [5.3515625] * 3), ([0.0, -0.0390625, 0.0], [9.3359375] * 3),
             ([0.1, -0.2, 0.3], [7.3, 7.3, 7.3]), ([-1.25, 0.5, 2.0], [3.0, 11.7, 6.1])]


def isv_query_points(center, diam, res, rng, n=4000):
    """What the mesher asks for (octree corners: dyadic fractions of the box, including the maximum corner = index res),
    lattice points themselves, points next to cell boundaries, and arbitrary points inside the box."""
    c, d = np.float32(center), np.float32(diam)
    lo = c - d * np.float32(0.5)
    frac = rng.integers(0, res + 1, size=(n, 3)).astype(np.float32) / np.float32(res)
    pts = [lo + d * frac,
           lo + d * rng.random((n, 3), dtype=np.float32),
           np.nextafter(lo + d * frac, np.float32(np.inf)), np.nextafter(lo + d * frac, np.float32(-np.inf)),
           np.array([c + d * np.float32(0.5), c - d * np.float32(0.5), c], dtype=np.float32)]
    return np.ascontiguousarray(np.concatenate(pts).astype(np.float32))


@pytest.mark.parametrize("res", [128, 512, 1024])
def test_lattice_snap_and_positions_equal_the_reference_isv(ref, oracle, res):
    """ISV3D64::getCoords (ISV.hpp:91-96, the truncating snap every mesher sample goes through) and getPoint
    (:103-109), bit for bit: the oracle's oracle_lattice_coords / oracle_lattice_point against the reference's class."""
    rng = np.random.default_rng(res)
    for center, diam in ISV_BOXES:
        pts = isv_query_points(center, diam, res, rng)
        want = np.zeros((len(pts), 3), dtype=np.int64)
        cf, df = np.float32(center), np.float32(diam)
        ref.ref_isv_coords(p(cf), p(df), res, 16, p(pts), len(pts), p(want))
        got = np.array([oracle.lattice_coords(center, diam, res, q) for q in pts[::7]])
        assert np.array_equal(got, want[::7])
        assert want.min() >= -1 and want.max() <= res + 1 and (want == res).any()   # index res is reachable (quirk 5)
        idx = np.ascontiguousarray(np.stack([np.arange(res + 1)] * 3, axis=1).astype(np.int32))
        idx[:, 1] = idx[::-1, 1]
        want_p = np.zeros((res + 1, 3), dtype=np.float32)
        ref.ref_isv_points(p(cf), p(df), res, p(idx), len(idx), p(want_p))
        got_p = np.array([oracle.lattice_point(center, diam, res, *map(int, t)) for t in idx])
        assert np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32))


def test_lattice_value_is_the_value_at_the_snapped_point(ref, oracle):
    """ISV3D64::getValue (ISV.hpp:15-60) returns the data source's value AT THE SNAPPED LATTICE POINT -- block key
    (:85-87), block point order (:28-33), in-block index (:97-102, quirk 7), the block beyond the box for index res
    (quirk 5), eviction and re-fetch (:110-124) included.  The oracle's sampler is 'evaluate at
    lattice_point(clamp(lattice_coords(p)))': both the point and the value must agree bit for bit."""
    res, sub = 128, 16
    rng = np.random.default_rng(5)
    f = lambda q: (q[:, 0] * np.float32(1.25) + q[:, 1] * np.float32(-0.75)) + q[:, 2] * np.float32(3.5)
    calls = []

    @ISV_EVAL
    def source(xyz, n, out, user):
        q = np.ctypeslib.as_array(xyz, shape=(n, 3))
        np.ctypeslib.as_array(out, shape=(n,))[:] = f(q)
        calls.append(int(n))

    for center, diam in ISV_BOXES[:3]:
        pts = isv_query_points(center, diam, res, rng, n=150)
        # only points the mesher can produce: inside the closed box (coords 0..res)
        cf, df = np.float32(center), np.float32(diam)
        co = np.zeros((len(pts), 3), dtype=np.int64)
        ref.ref_isv_coords(p(cf), p(df), res, sub, p(pts), len(pts), p(co))
        pts = np.ascontiguousarray(pts[((co >= 0) & (co <= res)).all(1)])
        for gc in (-1, 8):   # the export's "never collect" and an aggressive collector: the values cannot depend on it
            del calls[:]
            values = np.zeros(len(pts), dtype=np.float32)
            at = np.zeros((len(pts), 3), dtype=np.float32)
            blocks = ref.ref_isv_values(p(cf), p(df), res, sub, 16, gc, p(pts), len(pts), source, None, p(values), p(at))
            assert blocks == len(calls) and all(n == (res // sub) ** 3 for n in calls)
            want_at = np.array([oracle.lattice_point(center, diam, res,
                                                     *np.clip(oracle.lattice_coords(center, diam, res, q), 0, res))
                                for q in pts])
            assert np.array_equal(at.view(np.uint32), want_at.view(np.uint32))
            assert np.array_equal(values.view(np.uint32), f(want_at).astype(np.float32).view(np.uint32))


def test_writers_equal_the_reference_writers(ref, oracle, built_lib, tmp_path):
    """cms::writeTrianglesToSTL / writeTrianglesToPLY (utils.hpp:41-154, through happly.h) byte for byte: the
    oracle's writers and the product's parallel writers."""
    from designcsg_amd.evaluator import write_ply, write_stl
    rng = np.random.default_rng(11)
    for n in (0, 1, 7, 40_000):
        tris = rng.normal(scale=3.0, size=(n, 3, 3)).astype(np.float32)
        if n > 2:
            tris[1, 0] = (np.float32(-0.0), np.float32(1e-40), np.float32(3.4e38))      # -0, denormal, huge
        assert ref.ref_write_stl(str(tmp_path / "r.stl").encode(), p(tris), n) == n
        assert ref.ref_write_ply(str(tmp_path / "r.ply").encode(), p(tris), n) == n
        oracle.write_stl(str(tmp_path / "o.stl"), tris)
        oracle.write_ply(str(tmp_path / "o.ply"), tris)
        write_stl(tmp_path / "p.stl", tris)
        write_ply(tmp_path / "p.ply", tris)
        want_stl, want_ply = (tmp_path / "r.stl").read_bytes(), (tmp_path / "r.ply").read_bytes()
        assert (tmp_path / "o.stl").read_bytes() == want_stl and (tmp_path / "p.stl").read_bytes() == want_stl
        assert (tmp_path / "o.ply").read_bytes() == want_ply and (tmp_path / "p.ply").read_bytes() == want_ply


def test_lookup_table_expansion_equals_the_reference(ref, oracle, built_lib):
    """cms::getIndexTrianglesFromTable + getIndexTriangleStrip (readLookupTable.hpp:32-76, geometry.hpp:228-248)."""
    count = np.zeros(256, dtype=np.int32)
    tris = np.zeros((256, 5, 3), dtype=np.int32)
    n_cases = ref.ref_lookup_table(TABLE.encode(), p(count), p(tris))
    assert n_cases > 0
    mine = oracle.LookupTable(open(TABLE).read())
    assert np.array_equal(mine.count, count)
    for c in range(256):
        assert np.array_equal(mine.tris[c, :count[c]], tris[c, :count[c]]), c
    # the product's host parser (csrc/scene.cpp parse_lookup_table): table[case*16] = count, then 3 edge ids / triangle
    out = (C.c_ubyte * 4096)()
    total = C.c_int(0)
    built_lib.dcsg_expand_lookup_table.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_int)]
    assert built_lib.dcsg_expand_lookup_table(open(TABLE, "rb").read(), out, C.byref(total)) == 0
    t = np.frombuffer(out, dtype=np.uint8).reshape(256, 16)
    assert total.value == int(count.sum())
    for c in range(256):
        assert t[c, 0] == count[c] and np.array_equal(t[c, 1:1 + 3 * count[c]].reshape(-1, 3), tris[c, :count[c]]), c


def test_corner_and_child_conventions_equal_the_reference(ref):
    """Box3f::getCorners (geometry.hpp:264-279) and Node::subdivideIntoQueue (octree.hpp:24-32), in float, against
    the expressions the product and the oracle use (corner = c + s*(hd*sign), child centre = corner(0.5),
    child hd = 0.5*hd) with the sign table of DESIGN.md."""
    SG = np.array([[-1, -1, 1], [1, -1, 1], [1, -1, -1], [-1, -1, -1], [-1, 1, 1], [1, 1, 1], [1, 1, -1], [-1, 1, -1]],
                  dtype=np.float32)
    rng = np.random.default_rng(2)
    for _ in range(200):
        c = rng.normal(scale=3, size=3).astype(np.float32)
        hd = np.abs(rng.normal(scale=2, size=3)).astype(np.float32) + np.float32(1e-3)
        for s in (np.float32(1.0), np.float32(0.5)):
            got = np.zeros((8, 3), dtype=np.float32)
            ref.ref_box_corners(p(c), p(hd), C.c_float(float(s)), p(got))
            want = (c[None, :] + s * (hd[None, :] * SG)).astype(np.float32)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        cc, ch = np.zeros((8, 3), dtype=np.float32), np.zeros(3, dtype=np.float32)
        ref.ref_subdivide(p(c), p(hd), p(cc), p(ch))
        want_c = (c[None, :] + np.float32(0.5) * (hd[None, :] * SG)).astype(np.float32)
        assert np.array_equal(cc.view(np.uint32), want_c.view(np.uint32))
        assert np.array_equal(ch.view(np.uint32), (np.float32(0.5) * hd).view(np.uint32))


def test_angle_test_folding_equals_the_reference(ref, built_lib):
    """mesh.hpp:253 compares Vector3f::angleBetweenVectors(a, b, 1e-6) (geometry.hpp:118-124: acosf of dot / (|a||b|),
    0 below the tolerance) with complexSurfaceThreshold; the product compares q = dot/(|a||b|) with a host-computed
    q_star.  Both decisions on random and adversarial vector pairs, for several thresholds.  (The reference calls
    libm's acosf; the fold uses (float)acos((double)q): equal except on double-rounding ties, which these samples
    would expose as a mismatch.)"""
    built_lib.dcsg_angle_q_star.restype = C.c_float
    built_lib.dcsg_angle_q_star.argtypes = [C.c_float]
    rng = np.random.default_rng(8)
    a = rng.normal(size=(20000, 3)).astype(np.float32)
    b = rng.normal(size=(20000, 3)).astype(np.float32)
    b[:5000] = a[:5000] + rng.normal(scale=0.3, size=(5000, 3)).astype(np.float32)      # small angles
    b[5000:6000] = a[5000:6000]                                                          # angle 0
    b[6000:7000] = -a[6000:7000]                                                         # angle pi
    for thr in (np.float32(np.pi / 4), np.float32(0.3), np.float32(1.0), np.float32(np.pi / 2), np.float32(3.0)):
        q_star = built_lib.dcsg_angle_q_star(C.c_float(float(thr)))
        for i in range(len(a)):
            ang = ref.ref_angle_between(p(a[i]), p(b[i]), C.c_float(1e-6))
            ma, mb = np.float32(ref.ref_magnitude(p(a[i]))), np.float32(ref.ref_magnitude(p(b[i])))
            dot = np.float32(np.float32(np.float32(a[i, 0] * b[i, 0]) + np.float32(a[i, 1] * b[i, 1])) + np.float32(a[i, 2] * b[i, 2]))
            mm = np.float32(ma * mb)
            mine = (not (mm < np.float32(1e-6))) and (np.float32(-1.0) <= np.float32(dot / mm) <= np.float32(q_star))
            assert (ang > thr) == bool(mine), (i, float(thr), float(ang))
            if i > 3000 and thr != np.float32(np.pi / 4):
                break       # the full sample for the default threshold, a prefix for the others
