"""CPU-side checks of the product: the C-ABI library loads and exports exactly what include/dcsg.h
declares, fails loudly without a GPU, and its host logic (parsers, writers, lookup-table expansion,
build-procedure optimiser, brush JIT) agrees with the oracle.  No kernel is launched here."""
import ctypes as C
import os
import sys
import re

import numpy as np
import pytest

from conftest import ROOT, TABLE, read_scene, scene_dir

HEADER = os.path.join(ROOT, "include", "dcsg.h")
ALL_SCENES = ["sphere", "design1", "design2", "mixed_groups", "flat256", "grouped256", "ad_table", "prefabs", "const_table", "global_state", "userbrush128"]


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(dcsg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_lib):
    from designcsg_amd import _lib
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(built_lib, n), "libdcsg_hip.so does not export " + n
    assert sorted(_lib.PROTOTYPES) == names, "ctypes prototypes and include/dcsg.h disagree"


def test_header_is_plain_c_and_a_c_program_links(built_lib, tmp_path):
    """include/dcsg.h is the boundary a maintainer binds: it must compile as C99 and as C++, and a
    C program linked against libdcsg_hip.so must get the documented behaviour without a GPU (dcsg_create fails with
    DCSG_ERR_NO_DEVICE and a message; the host-only entry points work)."""
    import subprocess
    from designcsg_amd import build as b
    src = tmp_path / "client.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "dcsg.h"
int main(void) {
    dcsg_ctx* ctx = NULL;
    int rc = dcsg_create(&ctx, 0);
    dcsg_export_config cfg;
    int rc2 = dcsg_parse_export_config("10.0\n6\n8\n9\n0.785\n50\n16\n512\n4096\n", &cfg);
    float tri[9] = {0, 0, 0, 1, 0, 0, 0, 1, 0};
    int rc3 = dcsg_write_stl("one.stl", tri, 1);
    printf("%d|%s|%d|%d|%d|%d\n", rc, dcsg_last_error(NULL), rc2, cfg.grid_level, cfg.gradient_descent_steps, rc3);
    if (ctx) dcsg_destroy(ctx);
    return 0;
}
''')
    inc = os.path.join(ROOT, "include")
    for cc, std, lang in (("gcc", "-std=c99", "c"), ("g++", "-std=c++17", "c++")):
        subprocess.run([cc, std, "-x", lang, "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I" + inc, str(src)], check=True)
    exe = tmp_path / "client"
    subprocess.run(["gcc", "-std=c99", "-I" + inc, str(src), "-o", str(exe), "-L" + b.OUT, "-ldcsg_hip",
                    "-Wl,-rpath," + b.OUT, "-Wl,-rpath," + os.path.join(b.ROCM, "lib")], check=True)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([str(exe)], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rc, msg, rc2, grid, gd, rc3 = r.stdout.strip().split("|")
    assert int(rc) == -2 and "no CPU fallback" in msg          # DCSG_ERR_NO_DEVICE, loudly
    assert (int(rc2), int(grid), int(gd), int(rc3)) == (0, 9, 50, 0)
    assert os.path.getsize(tmp_path / "one.stl") == 84 + 50


def test_no_symbol_of_the_oracle_is_linked(built_lib):
    import subprocess
    from designcsg_amd import _lib
    out = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle_" not in out
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "libdcsg_oracle" not in needed and "libamdhip64" in needed


def test_product_sources_never_touch_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "designcsg_amd")):
        if "_build" in base:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", ".cl")):
                t = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|#include\s+[\"<].*oracle", t, re.M):
                    bad.append(f)
    assert not bad, bad


def test_create_fails_loudly_without_a_gpu(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from designcsg_amd import _lib
    from designcsg_amd.evaluator import Evaluator
    with pytest.raises(_lib.DcsgError) as ei:
        Evaluator(0)
    assert ei.value.code == _lib.ERR_NO_DEVICE and "no CPU fallback" in ei.value.message


def test_import_fails_loudly_when_the_library_is_missing(monkeypatch):
    from designcsg_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdcsg_hip.so")
    with pytest.raises(ImportError):
        _lib.lib()


def test_export_config_parser(built_lib, oracle):
    from designcsg_amd.evaluator import parse_export_config
    for name in ALL_SCENES:
        txt = read_scene(name)["exportConfig.txt"]
        a, b = parse_export_config(txt), oracle.parse_export_config(txt)
        for f, _ in a._fields_:
            assert getattr(a, f) == getattr(b, f), (name, f)
    a = parse_export_config("10.0\n6\n8\n9\n0.7853981633974483\n50\n16\n512\n4096\n")
    assert (a.bounding_box_half_diameter, a.min_level, a.max_level, a.grid_level) == (10.0, 6, 8, 9)
    assert a.complex_surface_threshold == np.float32(0.7853981633974483)
    with pytest.raises(ValueError):
        parse_export_config("10.0\n6\n8\n")


def test_opt_in_writers(built_lib, tmp_path):
    """Writers beyond the reference's: indexed PLY and STL with facet normals.  Checked against a
    numpy restatement: the indexed file expands back to the same triangle soup, vertices appear once in order of
    first appearance, normals are normalize(cross(B-A, C-A)) of the vertices as written, in float32."""
    from designcsg_amd.evaluator import write_ply_indexed, write_stl, write_stl_normals
    rng = np.random.default_rng(5)
    for n in (0, 1, 5000, 40000):   # the last one is above the single-thread threshold of the writers
        grid = rng.normal(size=(max(n // 3, 4), 3)).astype(np.float32)     # shared pool => shared vertices
        tris = grid[rng.integers(0, len(grid), (n, 3))] if n else np.zeros((0, 3, 3), np.float32)
        if n > 2:
            tris[1] = tris[0, 0]                                            # degenerate facet: zero normal stays
            tris[2, 1] = np.float32(-0.0) * tris[2, 1]                      # -0.0 and +0.0 are different vertices
        nv = write_ply_indexed(tmp_path / "i.ply", tris)
        raw = (tmp_path / "i.ply").read_bytes()
        head, body = raw[:raw.index(b"end_header\n") + 11], raw[raw.index(b"end_header\n") + 11:]
        assert b"element vertex %d\n" % nv in head and b"element face %d\n" % n in head and b"property float x" in head
        v = np.frombuffer(body[:12 * nv], dtype="<f4").reshape(-1, 3)
        f = np.frombuffer(body[12 * nv:], dtype=np.uint8).reshape(-1, 13)
        assert len(f) == n and (f[:, 0] == 3).all()
        idx = f[:, 1:].copy().view("<u4").reshape(-1, 3)
        assert np.array_equal(v[idx].view(np.uint32), tris.view(np.uint32).reshape(-1, 3, 3))     # expands to the soup
        flat = tris.reshape(-1, 3).view(np.uint32)
        _, first = np.unique(flat, axis=0, return_index=True)
        assert nv == len(first) and np.array_equal(v.view(np.uint32), flat[np.sort(first)])        # once, first-seen order
        # STL with normals: same bytes as the plain writer except the 12 normal bytes of each record
        write_stl(tmp_path / "p.stl", tris)
        write_stl_normals(tmp_path / "n.stl", tris)
        a = np.frombuffer((tmp_path / "p.stl").read_bytes()[84:], dtype=np.uint8).reshape(-1, 50)
        b = np.frombuffer((tmp_path / "n.stl").read_bytes()[84:], dtype=np.uint8).reshape(-1, 50)
        assert (tmp_path / "p.stl").read_bytes()[:84] == (tmp_path / "n.stl").read_bytes()[:84]
        assert np.array_equal(a[:, 12:], b[:, 12:])
        if n:
            w = tris[:, :, [0, 2, 1]]                                       # as written: (x, z, y)
            u, t = w[:, 1] - w[:, 0], w[:, 2] - w[:, 0]
            c = np.stack([u[:, 1] * t[:, 2] - u[:, 2] * t[:, 1], u[:, 2] * t[:, 0] - u[:, 0] * t[:, 2],
                          u[:, 0] * t[:, 1] - u[:, 1] * t[:, 0]], axis=1).astype(np.float32)
            ln = np.sqrt((c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]) + c[:, 2] * c[:, 2]).astype(np.float32)
            want = np.where(ln[:, None] > 0, c / np.where(ln > 0, ln, 1)[:, None], 0).astype(np.float32)
            got = b[:, :12].copy().view("<f4")
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
            if n > 2:
                assert (got[1] == 0).all() and np.allclose(np.linalg.norm(got[0]), 1, atol=1e-6)


def test_writers_byte_identical_to_oracle(built_lib, oracle, tmp_path):
    from designcsg_amd.evaluator import write_ply, write_stl
    rng = np.random.default_rng(0)
    for n in (0, 1, 1000, 3, 0):   # the same paths again and again: files are overwritten in place, then cut
        tris = rng.normal(size=(n, 3, 3)).astype(np.float32)
        write_stl(tmp_path / "a.stl", tris)
        write_ply(tmp_path / "a.ply", tris)
        oracle.write_stl(str(tmp_path / "b.stl"), tris)
        oracle.write_ply(str(tmp_path / "b.ply"), tris)
        stl, ply = (tmp_path / "a.stl").read_bytes(), (tmp_path / "a.ply").read_bytes()
        assert stl == (tmp_path / "b.stl").read_bytes()
        assert ply == (tmp_path / "b.ply").read_bytes()
        # utils.hpp:41-103: 80 zero bytes, count, 50-byte records, (x, z, y) order, zero normals
        assert len(stl) == 84 + 50 * n and stl[:80] == bytes(80) and int.from_bytes(stl[80:84], "little") == n
        if n:
            rec = np.frombuffer(stl[84:134], dtype=np.uint8)
            f = rec[:48].view(np.float32)
            assert (f[:3] == 0).all() and f[3] == tris[0, 0, 0] and f[4] == tris[0, 0, 2] and f[5] == tris[0, 0, 1]
        # utils.hpp:106-154 / happly: exact header, doubles, uchar+3*uint faces
        hdr = ("ply\nformat binary_little_endian 1.0\ncomment Written with hapPLY (https://github.com/nmwsharp/happly)\n"
               "element vertex %d\nproperty double x\nproperty double y\nproperty double z\n"
               "element face %d\nproperty list uchar uint vertex_indices\nend_header\n" % (3 * n, n)).encode()
        assert ply.startswith(hdr) and len(ply) == len(hdr) + 85 * n


def test_range_writers_and_crash_safety(built_lib, tmp_path):
    """dcsg_write_{stl,ply}_range (what the ranks of a sharded export call): any partition of the triangles, written in
    any order between BEGIN and END, gives the single-shot file byte for byte; between BEGIN and END the file does not
    look like a finished STL / PLY; a failed write leaves an empty file (never a valid header over a stale body)."""
    from designcsg_amd.evaluator import write_ply, write_stl
    rng = np.random.default_rng(3)
    n = 70000  # more than one writer buffer, more than one thread
    tris = rng.normal(size=(n, 3, 3)).astype(np.float32)
    write_stl(tmp_path / "whole.stl", tris)
    write_ply(tmp_path / "whole.ply", tris)
    cuts = [0, 1, 12345, 12345, 50000, n]  # includes an empty range
    order = [3, 0, 4, 2, 1]
    for ext, fn in (("stl", built_lib.dcsg_write_stl_range), ("ply", built_lib.dcsg_write_ply_range)):
        path = tmp_path / ("parts." + ext)
        # a longer stale file of the same name: the tail must go, and no reader may see old + new mixed as valid
        (tmp_path / ("whole." + ext)).read_bytes()
        path.write_bytes((tmp_path / ("whole." + ext)).read_bytes() + b"stale tail" * 1000)
        assert fn(os.fsencode(path), None, 0, 0, n, 1) == 0                      # BEGIN
        head = path.read_bytes()[:40]
        assert head.startswith(b"dcsg: incomplete export")
        for k in order:
            a, b = cuts[k], cuts[k + 1]
            part = np.ascontiguousarray(tris[a:b])
            assert fn(os.fsencode(path), part.ctypes.data_as(C.c_void_p) if b > a else None, b - a, a, n, 0) == 0
            assert path.read_bytes()[:23] == b"dcsg: incomplete export"            # still not a finished file
        assert fn(os.fsencode(path), None, 0, 0, n, 2) == 0                      # END
        assert path.read_bytes() == (tmp_path / ("whole." + ext)).read_bytes()
        # argument checks: range beyond the total; payload into a file that was never begun
        assert fn(os.fsencode(path), tris.ctypes.data_as(C.c_void_p), 10, n - 5, n, 0) != 0
        assert fn(os.fsencode(tmp_path / ("never." + ext)), tris.ctypes.data_as(C.c_void_p), 10, 0, n, 0) != 0
    # a write that fails part-way (file size limit) leaves an empty file behind
    code = (
        "import resource, signal, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from designcsg_amd.evaluator import write_stl, write_ply\n"
        "from designcsg_amd._lib import DcsgError\n"
        "signal.signal(signal.SIGXFSZ, signal.SIG_IGN)\n"
        "resource.setrlimit(resource.RLIMIT_FSIZE, (1 << 20, 1 << 20))\n"
        "t = np.ones((100000, 3, 3), dtype=np.float32)\n"
        "for f, w in ((%r, write_stl), (%r, write_ply)):\n"
        "    try:\n"
        "        w(f, t)\n"
        "        sys.exit(3)\n"
        "    except DcsgError:\n"
        "        pass\n" % (ROOT, str(tmp_path / "whole.stl"), str(tmp_path / "whole.ply")))
    import subprocess
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.getsize(tmp_path / "whole.stl") == 0 and os.path.getsize(tmp_path / "whole.ply") == 0


def test_lookup_table_expansion_matches_oracle(built_lib, lookup_table):
    table = np.zeros(4096, dtype=np.uint8)
    total = C.c_int()
    assert built_lib.dcsg_expand_lookup_table(open(TABLE, "rb").read(), table.ctypes.data_as(C.c_void_p),
                                              C.byref(total)) == 0
    assert total.value == lookup_table.total == 820
    t = table.reshape(256, 16)
    for case in range(256):
        n = int(lookup_table.count[case])
        assert t[case, 0] == n
        assert t[case, 1:1 + 3 * n].tolist() == lookup_table.tris[case, :n].reshape(-1).tolist()
    bad = np.zeros(4096, dtype=np.uint8)
    assert built_lib.dcsg_expand_lookup_table(b"0,3,8\n", bad.ctypes.data_as(C.c_void_p), None) != 0


# ---- build-procedure optimiser -------------------------------------------------------------------
def compile_program(lib, scene_txt, bp_txt, cap=4096):
    reg = np.zeros((cap, 4), dtype=np.int32)
    stk = np.zeros((cap, 4), dtype=np.int32)
    n_reg, n_regs, n_stk, depth = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib.dcsg_compile_program(scene_txt.encode(), bp_txt.encode(), reg.ctypes.data_as(C.c_void_p), C.byref(n_reg),
                                  C.byref(n_regs), stk.ctypes.data_as(C.c_void_p), C.byref(n_stk), C.byref(depth), cap)
    return rc, reg[:n_reg.value], n_regs.value, stk[:n_stk.value], depth.value


class object_values(dict):
    """(brush b, object i) -> sdf_bank(local(v, object i), b) from the oracle: the value an IMPORT yields."""

    def __init__(self, oracle, scene, pts):
        super().__init__()
        self.oracle, self.scene, self.pts = oracle, scene, pts
        self.cl = os.path.join(scene["dir"], "scene.cl")

    def __missing__(self, key):
        brush, obj = key
        mini = _oracle_scene(self.oracle, self.scene["scene.txt"], "0 %d %d 0\n1 0 -1 -1" % (brush, obj), self.scene)
        self[key] = mini.eval_sdf(self.pts)
        return self[key]


def _oracle_scene(oracle, scene_txt, bp, scene):
    ad = None
    if "arbitrary_data" in scene:
        ad = np.zeros(131072, dtype=np.float32)
        ad[:len(scene["arbitrary_data"])] = scene["arbitrary_data"]
    return oracle.Scene(scene_txt, bp, os.path.join(scene["dir"], "scene.cl"), ad)


def tmin(a, b):
    return np.where(a < b, a, b)


def tmax(a, b):
    return np.where(a > b, a, b)


def as_f32(bits):
    return np.array([bits], dtype=np.int32).view(np.float32)[0]


def smooth(a, b, k, inv_k, is_min):
    """Extension opcodes SMOOTH_MIN / SMOOTH_MAX restated in numpy float32, one rounding per operation."""
    t = k - np.abs(a - b)
    t = np.where(t > np.float32(0), t, np.float32(0)).astype(np.float32)
    h = t * inv_k
    q = ((h * h) * k) * np.float32(0.25)
    return (tmin(a, b) - q) if is_min else (tmax(a, b) + q)


def run_register_program(prog, n_regs, imp, n):
    st = np.zeros((max(n_regs, 1), n), dtype=np.float32)
    out = np.full(n, 64.0, dtype=np.float32)
    k = inv_k = np.float32(0)
    for op, l, r, d in prog.tolist():
        if op == 8: k, inv_k = as_f32(l), as_f32(r)
        elif op == 6: st[d] = smooth(st[l], st[r], k, inv_k, True)
        elif op == 7: st[d] = smooth(st[l], st[r], k, inv_k, False)
        elif op == 19: st[d] = smooth(st[d], imp[(l & 255, r)], k, inv_k, True)
        elif op == 20: st[d] = smooth(st[d], imp[(l & 255, r)], k, inv_k, False)
        elif op == 21: st[d] = smooth(st[d], -imp[(l & 255, r)], k, inv_k, False)
        elif op == 0: st[d] = imp[(l & 255, r)]
        elif op == 16: st[d] = tmin(st[d], imp[(l & 255, r)])
        elif op == 17: st[d] = tmax(st[d], imp[(l & 255, r)])
        elif op == 18: st[d] = tmax(st[d], -imp[(l & 255, r)])
        elif op == 1: out = st[l].copy()
        elif op == 2: st[d] = tmin(st[l], st[r])
        elif op == 3: st[d] = tmax(st[l], st[r])
        elif op == 4: st[d] = -st[l]
        elif op == 5: st[d] = st[l].copy()
    return out


def run_stack_program(prog, imp, n):
    st = []
    out = np.full(n, 64.0, dtype=np.float32)
    k = inv_k = np.float32(0)
    for op, l, r, depth in prog.tolist():
        if op == 40:
            k, inv_k = as_f32(l), as_f32(r)
            continue
        assert depth == len(st), "recorded depth must equal the stack depth before the step"
        if op == 41: st[-1] = smooth(st[-1], imp[(l & 255, r)], k, inv_k, True)
        elif op == 42: st[-1] = smooth(st[-1], imp[(l & 255, r)], k, inv_k, False)
        elif op == 43: st[-1] = smooth(st[-1], -imp[(l & 255, r)], k, inv_k, False)
        elif op in (44, 45, 46):
            v = st.pop()
            st[-1] = smooth(st[-1], v, k, inv_k, True) if op == 44 else smooth(st[-1], v if op == 45 else -v, k, inv_k, False)
        elif op == 32: st.append(imp[(l & 255, r)].copy())
        elif op == 33: st[-1] = tmin(st[-1], imp[(l & 255, r)])
        elif op == 34: st[-1] = tmax(st[-1], imp[(l & 255, r)])
        elif op == 35: st[-1] = tmax(st[-1], -imp[(l & 255, r)])
        elif op in (36, 37, 38):
            v = st.pop()
            st[-1] = tmin(st[-1], v) if op == 36 else (tmax(st[-1], v) if op == 37 else tmax(st[-1], -v))
        elif op == 39: out = st[-1].copy()
    return out


@pytest.mark.parametrize("name", ALL_SCENES)
def test_program_optimiser_preserves_values(built_lib, oracle, name):
    """The fused/renamed register program and the stack program, interpreted in numpy float32 over the
    oracle's per-object values, give the oracle's result for the ORIGINAL buildprocedure.txt, bit for bit."""
    sc = read_scene(name)
    sc["dir"] = scene_dir(name)
    rc, reg, n_regs, stk, depth = compile_program(built_lib, sc["scene.txt"], sc["buildprocedure.txt"])
    assert rc == 0
    n_orig = len(sc["buildprocedure.txt"].split("\n"))
    assert len(reg) < n_orig and n_regs <= 3          # fusion shortens, renaming needs root + group + R0 at most
    assert len(stk) > 0 and depth in (1, 2)           # compiler output is stack-disciplined
    rng = np.random.default_rng(4)
    pts = rng.uniform(-5, 5, (3000 if name in ("flat256", "grouped256") else 20000, 3)).astype(np.float32)
    want = oracle.Scene.from_dir(scene_dir(name)).eval_sdf(pts)
    imp = object_values(oracle, sc, pts)
    assert np.array_equal(run_register_program(reg, n_regs, imp, len(pts)).view(np.uint32), want.view(np.uint32))
    assert np.array_equal(run_stack_program(stk, imp, len(pts)).view(np.uint32), want.view(np.uint32))


def test_unknown_opcodes_are_dropped_before_their_operands_are_looked_at(built_lib):
    """k2.cl's switch ignores opcodes it does not know (k2.cl:59-140).  The numbers 16-21 and 32-46 are the HOST's own
    fused / stack opcodes: a file carrying them (with operands far outside every table) must compile to exactly the
    program of the same file without those lines -- nothing indexes with their slots or objects."""
    import ctypes as C
    scene_txt = read_scene("mixed_groups")["scene.txt"]
    clean = "0 0 0 0\n0 2 1 1\n2 0 1 0\n1 0 -1 -1"
    junk_ops = list(range(9, 64)) + [-1, -7, 255, 1000, 2 ** 31 - 1]
    dirty = ["0 0 0 0"]
    for i, op in enumerate(junk_ops):
        dirty.append("%d %d %d %d" % (op, 2, 5000 + i, (-1) ** i * (5000 + 17 * i)))
    dirty += ["0 2 1 1", "16 2 0 5000", "2 0 1 0", "35 4 99999 -12345", "1 0 -1 -1", "46 0 0 70000"]

    def compile_(bp):
        cap = 256
        reg, stk = (C.c_int32 * (4 * cap))(), (C.c_int32 * (4 * cap))()
        n_reg, n_regs, n_stk, depth = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rc = built_lib.dcsg_compile_program(scene_txt.encode(), bp.encode(), reg, C.byref(n_reg), C.byref(n_regs), stk,
                                            C.byref(n_stk), C.byref(depth), cap)
        assert rc == 0, built_lib.dcsg_last_error(None)
        return list(reg[:4 * n_reg.value]), n_regs.value, list(stk[:4 * n_stk.value]), depth.value

    assert compile_("\n".join(dirty)) == compile_(clean)
    # known opcodes with bad operands are still rejected, not ignored
    cap = 16
    buf = (C.c_int32 * (4 * cap))()
    z = C.c_int()
    for bad in ("0 2 1 64\n1 0 -1 -1", "0 2 999 0\n1 0 -1 -1", "2 0 77 0\n1 0 -1 -1", "1 -3 -1 -1"):
        assert built_lib.dcsg_compile_program(scene_txt.encode(), bad.encode(), buf, C.byref(z), C.byref(z), buf, C.byref(z),
                                              C.byref(z), cap) != 0, bad


def test_program_optimiser_on_hand_written_programs(built_lib, oracle):
    scene_txt = read_scene("mixed_groups")["scene.txt"]
    cases = {
        # R0 (slot 1) is read again later: no fusion of the first pair, and not stack-disciplined
        "live_r0": "0 2 1 1\n2 0 1 0\n0 3 2 0\n3 0 1 0\n1 0 -1 -1",
        # MIN with the operands the other way round must not be fused (T_min is not symmetric in bits)
        "swapped": "0 0 0 0\n0 2 1 1\n2 1 0 0\n1 0 -1 -1",
        # IDENTITY / NEGATE on their own
        "ident": "0 2 1 0\n5 0 -1 2\n4 2 -1 3\n3 0 3 0\n1 0 -1 -1",
        # a slot that is read before anything writes it (zero-initialised stack)
        "uninit": "0 2 1 1\n2 5 1 0\n1 0 -1 -1",
        # export of a slot other than the last one written
        "export_mid": "0 2 1 0\n0 3 2 1\n1 0 -1 -1",
        # unknown opcode is ignored (k2.cl's switch has no default)
        "unknown": "0 2 1 0\n9 0 0 0\n1 0 -1 -1",
        "empty": "",
    }
    rng = np.random.default_rng(9)
    pts = rng.uniform(-5, 5, (5000, 3)).astype(np.float32)
    sc = {"scene.txt": scene_txt, "dir": scene_dir("mixed_groups")}
    imp = object_values(oracle, sc, pts)
    for key, bp in cases.items():
        rc, reg, n_regs, stk, depth = compile_program(built_lib, scene_txt, bp)
        assert rc == 0, key
        want = oracle.Scene(scene_txt, bp, os.path.join(sc["dir"], "scene.cl")).eval_sdf(pts)
        got = run_register_program(reg, n_regs, imp, len(pts))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), key
        if len(stk):
            assert np.array_equal(run_stack_program(stk, imp, len(pts)).view(np.uint32), want.view(np.uint32)), key
    assert len(compile_program(built_lib, scene_txt, cases["live_r0"])[3]) == 0
    assert [s[0] for s in compile_program(built_lib, scene_txt, cases["swapped"])[1].tolist()] == [0, 0, 2, 1]


SMOOTH_SCENES = ["smooth_blobs", "smooth256"]   # extension fixtures (tests/golden/make_extension_fixtures.py)


@pytest.mark.parametrize("name", SMOOTH_SCENES)
def test_smooth_extension_programs(built_lib, oracle, name):
    """Smooth joins (extension opcodes 6-8, not in the reference).  Three independent statements of the same
    arithmetic must agree bit for bit: the oracle's interpreter on the original buildprocedure.txt, and the numpy
    float32 restatement above run on the host's fused register program and on its stack program."""
    sc = read_scene(name)
    sc["dir"] = scene_dir(name)
    rc, reg, n_regs, stk, depth = compile_program(built_lib, sc["scene.txt"], sc["buildprocedure.txt"])
    assert rc == 0 and len(stk) > 0 and depth in (2, 3)
    ops = set(stk[:, 0].tolist())
    assert 40 in ops and ops & {41, 42, 43}
    # one SET_K per group that needs one (every PUSH forgets the radius, so that groups are self-contained for the
    # tile culling), not one per join
    n_join = sum(1 for o in stk[:, 0].tolist() if 41 <= o <= 46)
    assert sum(1 for o in stk[:, 0].tolist() if o == 40) == {"smooth_blobs": 5, "smooth256": 16}[name] < n_join
    rng = np.random.default_rng(14)
    pts = rng.uniform(-5, 5, (3000 if name == "smooth256" else 20000, 3)).astype(np.float32)
    pts[: len(pts) // 2] *= np.float32(0.6)     # more samples inside the blend zones
    want = oracle.Scene.from_dir(scene_dir(name)).eval_sdf(pts)
    imp = object_values(oracle, sc, pts)
    assert np.array_equal(run_register_program(reg, n_regs, imp, len(pts)).view(np.uint32), want.view(np.uint32))
    assert np.array_equal(run_stack_program(stk, imp, len(pts)).view(np.uint32), want.view(np.uint32))
    # the blend actually happens somewhere: the smooth result differs from the hard one
    hard = sc["buildprocedure.txt"].replace("\n6 ", "\n2 ").replace("\n7 ", "\n3 ")
    ref = oracle.Scene(sc["scene.txt"], hard, os.path.join(sc["dir"], "scene.cl")).eval_sdf(pts)
    assert np.count_nonzero(ref != want) > len(pts) // 50


def test_smooth_extension_hand_written(built_lib, oracle):
    scene_txt = read_scene("mixed_groups")["scene.txt"]
    k1 = int(np.array([0.25], dtype=np.float32).view(np.int32)[0])
    k2 = int(np.array([1.5], dtype=np.float32).view(np.int32)[0])
    cases = {
        # unfused smooth ops on plain slots, radius changed in between, dest != left
        "plain": "0 2 1 0\n0 3 2 1\n8 %d 0 0\n6 0 1 2\n8 %d 0 0\n7 2 1 3\n6 3 0 0\n1 0 -1 -1" % (k1, k2),
        # operands swapped: must not be fused, same bits as the oracle
        "swapped": "0 4 3 0\n0 2 1 1\n8 %d 0 0\n6 1 0 0\n1 0 -1 -1" % k1,
        # leaf joins that fuse, incl. the carved (NEGATE + SMOOTH_MAX) form, then a hard join
        "fused": "0 0 0 0\n0 2 1 1\n8 %d 0 0\n6 0 1 0\n0 3 2 1\n4 1 -1 1\n8 %d 0 0\n7 0 1 0\n0 4 3 1\n2 0 1 0\n1 0 -1 -1" % (k1, k1),
    }
    rng = np.random.default_rng(19)
    pts = rng.uniform(-3, 3, (5000, 3)).astype(np.float32)
    sc = {"scene.txt": scene_txt, "dir": scene_dir("mixed_groups")}
    imp = object_values(oracle, sc, pts)
    for key, bp in cases.items():
        rc, reg, n_regs, stk, depth = compile_program(built_lib, scene_txt, bp)
        assert rc == 0, key
        want = oracle.Scene(scene_txt, bp, os.path.join(sc["dir"], "scene.cl")).eval_sdf(pts)
        assert np.array_equal(run_register_program(reg, n_regs, imp, len(pts)).view(np.uint32), want.view(np.uint32)), key
        if len(stk):
            assert np.array_equal(run_stack_program(stk, imp, len(pts)).view(np.uint32), want.view(np.uint32)), key
    assert [s[0] for s in compile_program(built_lib, scene_txt, cases["fused"])[1].tolist()] == [0, 8, 19, 21, 16, 1]
    # malformed: smooth join before any SET_K; radius zero / negative / NaN / inf
    for bad in ("0 2 1 0\n0 3 2 1\n6 0 1 0\n1 0 -1 -1", "8 0 0 0\n1 0 -1 -1", "8 -1082130432 0 0\n1 0 -1 -1",
                "8 2143289344 0 0\n1 0 -1 -1", "8 2139095040 0 0\n1 0 -1 -1"):
        assert compile_program(built_lib, scene_txt, bad)[0] != 0, bad


@pytest.mark.parametrize("seed", list(range(8)))
def test_program_optimiser_on_random_designs(built_lib, oracle, seed, tmp_path):
    """Generated designs (tests/designs/random_tree.py: every group kind, smooth joins, carved leaves, nesting): the
    host's fused register program and its stack program, run in numpy float32, equal the oracle's interpretation of
    the buildprocedure.txt the front-end wrote, bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "designs"))
    try:
        from random_tree import make_script
    finally:
        sys.path.pop(0)
    from designcsg_amd.run import run_design
    script = tmp_path / "design.py"
    script.write_text(make_script(2000 + seed, big=(seed % 2 == 1)))
    wd = run_design(str(script), str(tmp_path / "wd"))
    sc = {f: open(os.path.join(wd, f)).read() for f in ("scene.txt", "buildprocedure.txt", "scene.cl")}
    sc["dir"] = wd
    rc, reg, n_regs, stk, depth = compile_program(built_lib, sc["scene.txt"], sc["buildprocedure.txt"])
    assert rc == 0 and len(stk) > 0 and 1 <= depth <= 3
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-4, 4, (4000, 3)).astype(np.float32)
    want = oracle.Scene.from_dir(wd).eval_sdf(pts)
    imp = object_values(oracle, sc, pts)
    assert np.array_equal(run_register_program(reg, n_regs, imp, len(pts)).view(np.uint32), want.view(np.uint32))
    assert np.array_equal(run_stack_program(stk, imp, len(pts)).view(np.uint32), want.view(np.uint32))


def test_scene_validation(built_lib):
    scene_txt = read_scene("sphere")["scene.txt"]
    for bad in ("0 2 7 1\n1 0 -1 -1",       # object 7 of 2
                "0 2 1 64\n1 0 -1 -1",      # slot 64 (STACK_MEMORY_PER_PIXEL = 64)
                "0 2 1 -1\n1 0 -1 -1",      # negative destination
                "2 0 99 0\n1 0 -1 -1",
                "0 2 1\n"):                 # short line
        assert compile_program(built_lib, scene_txt, bad)[0] != 0, bad
    assert compile_program(built_lib, "0 1 0 0 0 0.2 0 0\n", "1 0 -1 -1")[0] != 0  # short scene.txt row


def test_brush_jit_without_a_gpu(built_lib, tmp_path, monkeypatch):
    """Evaluator::build's compile half: good source -> code object in the cache, bad source -> the log."""
    monkeypatch.setenv("DCSG_JIT_CACHE", str(tmp_path))
    src = read_scene("design2")["scene.cl"].encode()
    log = C.create_string_buffer(1 << 16)
    cached = C.c_int(-1)
    assert built_lib.dcsg_jit_prebuild(src, log, len(log), C.byref(cached)) == 0 and cached.value == 0
    objs = [f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]
    # a scene module is two code objects: the export part (what dcsg_export launches) and the rest (csrc/jit.hpp)
    assert len(objs) == 2 and all(open(tmp_path / o, "rb").read(4) == b"\x7fELF" for o in objs)
    assert built_lib.dcsg_jit_prebuild(src, log, len(log), C.byref(cached)) == 0 and cached.value == 1
    bad = src.replace(b"hilbert_sdf(v)", b"no_such_function(v)")
    assert built_lib.dcsg_jit_prebuild(bad, log, len(log), C.byref(cached)) == -4
    assert b"no_such_function" in log.value


def test_balanced_slab_cuts(built_lib):
    """dcsg_export_slab's cut planes from a stage-A histogram (no device: dcsg_debug_slab_cuts).  Work concentrated in the
    middle of the box (Design2-like), off-centre, in a single plane, and absent: the cuts tile 0 .. res, sit on node
    boundaries of the sweep level, give every slab at least one node layer, and balance the histogram."""
    import ctypes as C
    L = built_lib
    cd, half_neg, center_z, diameter = np.float32(10.0 / 256), np.float32(-10.0 / 512), 0.0, 9.3359375
    z = half_neg + cd * (np.arange(256) - 128)

    def cuts_for(hist, min_level, grid_level, world):
        h = np.ascontiguousarray(hist, dtype=np.uint32)
        out = np.zeros(world + 1, dtype=np.int32)
        rc = L.dcsg_debug_slab_cuts(h.ctypes.data_as(C.c_void_p), C.c_float(cd), C.c_float(half_neg), C.c_float(center_z), C.c_float(diameter),
                                    min_level, grid_level, world, out.ctypes.data_as(C.c_void_p))
        return rc, out

    def work(hist, cuts, res):
        lat = (z - (center_z - diameter / 2)) / (diameter / res)       # lattice plane of every probe plane
        return [float(hist[(lat >= a) & (lat < b)].sum()) for a, b in zip(cuts[:-1], cuts[1:])]

    middle = (2000 * np.exp(-(z / 1.2) ** 2)).astype(np.uint32)          # a blob in the middle of the box
    for world in (2, 4, 8):
        rc, cuts = cuts_for(middle, 7, 10, world)
        assert rc == 0 and cuts[0] == 0 and cuts[-1] == 1024 and np.all(np.diff(cuts) >= 8) and np.all(cuts % 8 == 0), cuts
        w = work(middle, cuts, 1024)
        assert max(w) < 1.35 * sum(w) / world, (world, cuts, w)
        equal = work(middle, np.arange(world + 1) * (1024 // world), 1024)
        if world > 2:
            assert max(equal) > 1.5 * sum(equal) / world          # what equal slabs would have done
    off = np.roll(middle, 60)                                            # the same blob off-centre
    rc, cuts = cuts_for(off, 7, 10, 8)
    assert rc == 0 and np.all(np.diff(cuts) >= 8) and max(work(off, cuts, 1024)) < 1.4 * off.sum() / 8
    spike = np.zeros(256, dtype=np.uint32)
    spike[100] = 5000                                                    # everything in one probe plane: still a valid tiling
    rc, cuts = cuts_for(spike, 7, 10, 8)
    assert rc == 0 and cuts[0] == 0 and cuts[-1] == 1024 and np.all(np.diff(cuts) >= 8)
    rc, cuts = cuts_for(np.zeros(256, dtype=np.uint32), 7, 10, 8)        # nothing near a surface: the equal slabs stay
    assert rc != 0 and cuts.tolist() == [128 * k for k in range(9)]
    rc, cuts = cuts_for(middle, 2, 6, 8)                                 # sweep level 2 has 4 node layers < 8 slabs: equal slabs
    assert rc != 0 and cuts.tolist() == [8 * k for k in range(9)]
    rc, cuts = cuts_for(middle, 3, 6, 8)                                 # exactly one node layer per slab
    assert cuts.tolist() == [8 * k for k in range(9)]
