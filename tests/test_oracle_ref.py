"""The oracle and the product's host code against the REAL reference, where the reference builds on its own.

oracle/_ref/libdcsg_ref.so (oracle/Makefile target `ref`, built by __graft_entry__.build() wherever /root/reference
exists; the built file travels to the GPU box) wraps the reference's stand-alone headers -- utils.hpp (both writers),
readLookupTable.hpp, geometry.hpp, octree.hpp -- without any stand-in.  Its mesher, ISV cache and Evaluator cannot be
built here (wx / Win32 headers, OpenCL runtime), so those parts of the oracle stay pinned by the survey's golden
vectors (tests/test_oracle_golden.py)."""
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
    return L


def p(a):
    return a.ctypes.data_as(C.c_void_p)


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
