"""GPU tests at BASELINE.json's full sizes (C2 sphere 512^3, C3 Design2 512^3, C4 Design2 1024^3, C5 256
primitives).  The oracle cannot re-run these inside the test budget, so they are checked against
tests/golden/big_configs.json (oracle runs, themselves pinned to SURVEY.md 8(c)) and through
size-independent properties: triangle SET hashes, per-level counts, vertex sums, radius of the projected
sphere, idempotence of the sweep, Z-slab union = whole."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT, export_config_text, read_scene

pytestmark = pytest.mark.gpu
BIG = json.load(open(os.path.join(ROOT, "tests", "golden", "big_configs.json")))


@pytest.fixture(scope="module")
def ev(built_lib):
    from designcsg_amd.evaluator import Evaluator
    e = Evaluator(0)
    yield e
    e.close()


def tri_set_hash(tris):
    b = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9).view(np.uint32)
    b = b[np.lexsort(b.T[::-1])]
    return hashlib.sha256(b.tobytes()).hexdigest()


@pytest.mark.parametrize("key", sorted(BIG))
def test_full_size_export_matches_golden(ev, key, tmp_path):
    g = BIG[key]
    s = read_scene(g["scene"])
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
    cfg_txt = export_config_text(g["scene"], levels=g["levels"], gd=0)
    st = ev.export(cfg_txt)  # mesher only
    assert list(st.box_center) == g["box_center"] and st.box_diameter == g["box_diameter"]
    assert st.interior_points == g["interior_points"]
    assert st.triangles == g["triangles"]  # bit-match of the triangle count (BASELINE config 2)
    assert {str(l): int(st.mesh.triangles[l]) for l in range(16) if st.mesh.triangles[l]} == g["per_level"]
    assert {str(l): int(st.mesh.culled[l]) for l in range(16) if st.mesh.culled[l]} == g["culled"]
    assert {str(l): int(st.mesh.subdivided[l]) for l in range(16) if st.mesh.subdivided[l]} == g["subdivided"]
    assert tri_set_hash(ev.download_triangles()) == g["pre_gd_set_sha256"]
    # projection (50 steps) and the writers
    ev.project_triangles(50)
    got = ev.download_triangles()
    assert tri_set_hash(got) == g["gd50_set_sha256"]
    v = got.reshape(-1, 3).astype(np.float64)
    assert int(np.isnan(v).sum()) == g["gd50_nan"] == 0
    assert np.allclose(v.sum(0), g["gd50_sums"], rtol=0, atol=1e-3)
    assert abs(np.linalg.norm(v, axis=1).sum() - g["gd50_sum_abs"]) < 1e-3
    if g["scene"] == "sphere":
        r = np.linalg.norm(v, axis=1)
        assert abs(r.mean() - 2.5) < 1e-6 and np.abs(r - 2.5).max() < 1e-5  # on the surface to 1e-5
    if key == "design2_6_8_9":  # BASELINE config 3: files of the documented sizes
        from designcsg_amd.evaluator import write_ply, write_stl
        write_stl(tmp_path / "o.stl", got)
        write_ply(tmp_path / "o.ply", got)
        assert os.path.getsize(tmp_path / "o.stl") == 84 + 50 * len(got) == 19783484
        assert os.path.getsize(tmp_path / "o.ply") == 249 + 85 * len(got) == 33632029


@pytest.mark.parametrize("name,levels", [("design2", (6, 8, 9)), ("flat256", (6, 8, 9)), ("flat256", (5, 7, 9)),
                                         ("smooth256", (6, 8, 9)), ("grouped256", (5, 8, 9))])
def test_node_wise_sweep_equals_dense_sweep_at_512(ev, name, levels):
    """dcsg_export sweeps the lattice only inside the octree nodes that are alive at level max(min,2) (default);
    sweeping every lattice point instead must give the same triangles, the same per-level statistics, and evaluate
    several times more points.  Nodes of 9^3 and 17^3 points; specialised, culling and smooth-join kernels."""
    s = read_scene(name)
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
    cfg_txt = export_config_text(name, levels=list(levels), gd=3)
    a = ev.export(cfg_txt)
    ta = tri_set_hash(ev.download_triangles())
    b = ev.export(cfg_txt, dense_sweep=True)
    tb = tri_set_hash(ev.download_triangles())
    assert a.triangles == b.triangles and ta == tb
    for f in ("culled", "subdivided", "leaves", "triangles"):
        assert list(getattr(a.mesh, f)) == list(getattr(b.mesh, f)), f
    assert b.swept_nodes == 0 and b.swept_points == b.lattice_points == 513 ** 3
    m = levels[0]
    side = (1 << (9 - m)) + 1
    # of a swept node's side^3 points the sweep writes those the mesher can read (kernels.hip, dcsg_in_sweep_pattern): corners
    # and centres of nodes down to the maximum level, every point along the edges of the levels that make the edge test
    step, S = 1 << (9 - levels[1]), side - 1
    edges = levels[1] > m
    r = np.arange(side)
    lz, ly, lx = np.meshgrid(r, r, r, indexing="ij")
    if step > 1:
        corner = ((lx | ly | lz) & (step - 1)) == 0
        centre = ((lx & (step - 1)) == step // 2) & ((ly & (step - 1)) == step // 2) & ((lz & (step - 1)) == step // 2)
        e = 2 * step - 1
        on_edge = (((lx & e) == 0).astype(int) + ((ly & e) == 0) + ((lz & e) == 0) >= 2) & edges
        per_node = int((corner | centre | on_edge).sum())
        assert per_node < side ** 3
    else:
        per_node = side ** 3
    assert a.swept_nodes == 8 * a.mesh.subdivided[m - 1] - a.mesh.culled[m] and a.swept_points == a.swept_nodes * per_node
    assert a.swept_points * 3 < b.swept_points


def test_sphere_at_2048_cubed(ev):
    """One size beyond BASELINE.json's largest (grid level 11: a 2049^3 lattice, 8.6 G points, 34 GB if dense).  No
    oracle run exists at this size; the domain's own invariants do: every projected vertex lies on the sphere of
    radius 2.5 to 1e-5, the triangle count scales with the area (4x the 1024^3 count to within 1 %), and the
    node-wise sweep touches a fraction of the lattice."""
    s = read_scene("sphere")
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
    st = ev.export(export_config_text("sphere", levels=[8, 10, 11], gd=50))
    assert st.lattice_points == 2049 ** 3 and st.box_diameter == 5.3515625
    assert 0 < st.swept_points < st.lattice_points // 5
    t10 = ev.export(export_config_text("sphere", levels=[7, 9, 10], gd=0)).triangles
    assert abs(st.triangles / (4.0 * t10) - 1.0) < 0.01
    st = ev.export(export_config_text("sphere", levels=[8, 10, 11], gd=50))
    v = ev.download_triangles().reshape(-1, 3).astype(np.float64)
    r = np.linalg.norm(v, axis=1)
    assert len(v) == 3 * st.triangles and abs(r.mean() - 2.5) < 1e-6 and np.abs(r - 2.5).max() < 1e-5


def test_dense_and_on_demand_normals_agree_at_512(ev):
    s = read_scene("design2")
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
    cfg_txt = export_config_text("design2", levels=(6, 8, 9), gd=0)
    a = ev.export(cfg_txt, dense_normals=False)
    ha = tri_set_hash(ev.download_triangles())
    b = ev.export(cfg_txt, dense_normals=True)
    assert a.triangles == b.triangles == 395668 and ha == tri_set_hash(ev.download_triangles())


def test_sweep_is_idempotent_and_slab_consistent_at_512(ev):
    import torch
    s = read_scene("sphere")
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
    c, d, res = [0.0, 0.0, 0.0], 5.3515625, 512
    n1 = res + 1
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.empty(n1 ** 3, dtype=torch.float32, device="cuda")
    na = torch.empty(3 * n1 ** 3, dtype=torch.float32, device="cuda")
    ev.sweep_lattice_device(c, d, res, 0, n1, a.data_ptr(), na.data_ptr(), stream)
    b = torch.empty(n1 ** 3, dtype=torch.float32, device="cuda")
    ev.sweep_lattice_device(c, d, res, 0, n1, b.data_ptr(), 0, stream)  # SDF-only kernel path
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # 8 Z slabs written separately give the same lattice
    nb = torch.empty(3 * n1 ** 3, dtype=torch.float32, device="cuda")
    for r in range(8):
        lo, hi = r * 64, (r + 1) * 64 + (1 if r == 7 else 0)
        ev.sweep_lattice_device(c, d, res, lo, hi, b.data_ptr() + lo * n1 * n1 * 4, nb.data_ptr() + lo * n1 * n1 * 12, stream)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and torch.equal(na.view(torch.int32), nb.view(torch.int32))
    # analytic properties of the sphere field: |n| = 1 (or 0 at the centre), sdf = |p|/5 - 0.5 to rounding
    n = na.view(-1, 3)
    ln = n.norm(dim=1)
    assert float(((ln - 1).abs() < 1e-6).logical_or(ln == 0).float().mean()) == 1.0
    assert int((a < 0).sum()) == int((b < 0).sum()) > 0


def test_sharded_driver_single_rank_equals_export(ev):
    from designcsg_amd.evaluator import parse_export_config
    from designcsg_amd.sharded import HipBackend, sharded_export
    s = read_scene("design2")
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
    cfg_txt = export_config_text("design2", levels=(5, 7, 8), gd=8)
    st = ev.export(cfg_txt)
    want = tri_set_hash(ev.download_triangles())
    res = sharded_export(HipBackend(ev, "cuda:0"), parse_export_config(cfg_txt), 0, 1)
    assert res["counts"] == [st.triangles] and tri_set_hash(res["triangles"]) == want


@pytest.mark.parametrize("name,levels,world", [("design2", (6, 8, 9), 4), ("design2", (7, 9, 10), 8), ("flat256", (6, 8, 9), 2),
                                               ("sphere", (5, 7, 9), 8), ("design2", (2, 5, 6), 8)])
def test_slab_exports_write_the_whole_mesh_into_one_file(ev, name, levels, world, tmp_path):
    """dcsg_export_slab + dcsg_write_slab, the entry points every rank of a sharded export calls (here: the ranks one
    after the other on this GPU): node-wise sweep restricted to the slab, per-rank byte ranges of one .stl / .ply.
    The files hold exactly the single-GPU export's triangles (as a set; slab order on disk), the per-level statistics
    add up, and each rank sweeps only its share of the lattice.  (2,5,6) with 8 ranks: the sweep level (2) is coarser
    than the slabs, so the ranks fall back to dense slab sweeps.)"""
    from designcsg_amd.evaluator import WRITE_BEGIN, WRITE_END, write_range
    from designcsg_amd.sharded import read_ply_triangles, read_stl_triangles
    s = read_scene(name)
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
    cfg_txt = export_config_text(name, levels=list(levels), gd=6)
    whole = ev.export(cfg_txt, str(tmp_path / "w.stl"), str(tmp_path / "w.ply"))
    want = tri_set_hash(ev.download_triangles())
    stats = [ev.export_slab(cfg_txt, r, world) for r in range(world)]
    counts = [int(st.triangles) for st in stats]
    assert sum(counts) == whole.triangles
    # the cut planes: work-balanced by default (stage A's count of probes near the surface per z plane, the same integers on
    # every rank), on node boundaries of the sweep level, tiling 0 .. res without gaps
    res, node = 1 << levels[2], (1 << levels[2]) >> min(max(levels[0], 2), levels[2])
    assert stats[0].slab_z_lo == 0 and stats[-1].slab_z_hi == res
    assert all(stats[r].slab_z_hi == stats[r + 1].slab_z_lo for r in range(world - 1))
    assert all(st.slab_z_hi > st.slab_z_lo for st in stats)
    if (1 << min(max(levels[0], 2), levels[2])) >= world:     # otherwise (sweep level coarser than the slabs) the equal slabs stay
        assert all(st.slab_z_lo % node == 0 for st in stats)
    equal = [ev.export_slab(cfg_txt, r, world, equal_slabs=True) for r in range(world)]
    assert [(st.slab_z_lo, st.slab_z_hi) for st in equal] == [(r * res // world, (r + 1) * res // world) for r in range(world)]
    assert sum(int(st.triangles) for st in equal) == whole.triangles
    if (name, levels, world) == ("design2", (7, 9, 10), 8):
        # Design2 sits in the middle of its box: an equal eighth holds 1.77 x the mean, the balanced cuts stay under 1.3 x
        assert max(int(st.triangles) for st in equal) > 1.7 * whole.triangles / world
        assert max(counts) < 1.3 * whole.triangles / world, counts
    for f in ("culled", "subdivided", "leaves", "triangles"):
        assert [sum(int(getattr(st.mesh, f)[l]) for st in stats) for l in range(16)] == [int(v) for v in getattr(whole.mesh, f)], f
    if whole.swept_nodes and (1 << max(levels[0], 2)) >= world:
        assert sum(int(st.swept_nodes) for st in stats) == whole.swept_nodes
        assert max(int(st.swept_points) for st in stats) < whole.swept_points
    stl, ply = str(tmp_path / "s.stl"), str(tmp_path / "s.ply")
    total = sum(counts)
    for path in (stl, ply):
        write_range(path, None, 0, total, WRITE_BEGIN)
    for r in reversed(range(world)):           # any order: every rank owns its byte range
        ev.export_slab(cfg_txt, r, world)
        ev.write_slab(stl, ply, sum(counts[:r]), total)
    for path in (stl, ply):
        write_range(path, None, 0, total, WRITE_END)
    assert os.path.getsize(stl) == os.path.getsize(tmp_path / "w.stl") and os.path.getsize(ply) == os.path.getsize(tmp_path / "w.ply")
    tris = read_stl_triangles(stl)
    assert tri_set_hash(tris) == want == tri_set_hash(read_stl_triangles(tmp_path / "w.stl"))
    pv, pf = read_ply_triangles(ply)
    assert np.array_equal(pv.astype(np.float32).view(np.uint32), tris.view(np.uint32))
    assert np.array_equal(pf, np.arange(3 * total, dtype=np.uint32).reshape(-1, 3))


@pytest.mark.parametrize("name,levels,world", [("design2", (2, 5, 6), 2), ("design2", (2, 5, 6), 4), ("design2", (3, 5, 7), 8),
                                               ("flat256", (2, 5, 6), 4), ("sphere", (3, 6, 7), 8)])
def test_slab_exports_stay_inside_their_slab_buffers(built_lib, name, levels, world, monkeypatch):
    """Low minimum levels: the node list of the sweep level holds nodes outside the rank's slab (all 64 roots at level 2,
    every child of a parent that straddles slabs at level 3 with 8 slabs).  On a FRESH context -- one that never held
    the whole lattice, so its buffers are slab-sized -- with guard bands around the slab buffers
    (DCSG_DEBUG_LATTICE_GUARD): the node-wise sweep must leave the guards untouched (the export fails otherwise) and the
    slabs must add up to the single-GPU mesh."""
    from designcsg_amd.evaluator import Evaluator
    s = read_scene(name)
    cfg_txt = export_config_text(name, levels=list(levels), gd=4)
    whole_ev = Evaluator(0)
    try:
        assert whole_ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
        whole = whole_ev.export(cfg_txt)
        want = tri_set_hash(whole_ev.download_triangles())
    finally:
        whole_ev.close()
    monkeypatch.setenv("DCSG_DEBUG_LATTICE_GUARD", str(8 << 20))
    parts, stats = [], []
    for r in range(world):
        e = Evaluator(0)   # fresh: buffers sized by this slab alone
        try:
            assert e.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
            st = e.export_slab(cfg_txt, r, world)
            stats.append(st)
            parts.append(e.download_triangles().reshape(-1, 9))
        finally:
            e.close()
    assert sum(int(st.triangles) for st in stats) == whole.triangles
    assert tri_set_hash(np.concatenate(parts)) == want
    if (1 << max(levels[0], 2)) >= world:   # node-wise on every rank: each sweeps its own nodes only
        assert all(int(st.swept_nodes) > 0 or int(st.triangles) == 0 for st in stats)
        assert sum(int(st.swept_nodes) for st in stats) == whole.swept_nodes


def test_export_multi_through_the_c_boundary(ev, tmp_path):
    """dcsg_export_multi: n contexts in one process (here all on this GPU; one per GPU on a node), one host thread and
    one Z slab each, per-slab byte ranges of the two files.  The files hold the single-context export's triangle set."""
    from designcsg_amd.evaluator import Evaluator, export_multi
    from designcsg_amd.sharded import read_ply_triangles, read_stl_triangles
    s = read_scene("design2")
    cfg_txt = export_config_text("design2", levels=(6, 8, 9), gd=10)
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
    whole = ev.export(cfg_txt)
    want = tri_set_hash(ev.download_triangles())
    evs = [Evaluator(0) for _ in range(4)]
    try:
        for e in evs:
            assert e.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
        for _ in range(2):   # the second run overwrites the files in place
            st = export_multi(evs, cfg_txt, tmp_path / "m.stl", tmp_path / "m.ply")
            assert sum(int(x.triangles) for x in st) == whole.triangles == 395668
            tris = read_stl_triangles(tmp_path / "m.stl")
            assert tri_set_hash(tris) == want
            pv, _ = read_ply_triangles(tmp_path / "m.ply")
            assert np.array_equal(pv.astype(np.float32).view(np.uint32), tris.view(np.uint32))
            assert os.path.getsize(tmp_path / "m.stl") == 19783484 and os.path.getsize(tmp_path / "m.ply") == 33632029
    finally:
        for e in evs:
            e.close()


@pytest.mark.parametrize("name,levels", [("design2", (7, 9, 10)), ("sphere", (6, 8, 9))])
def test_pipelined_export_equals_sequential(ev, name, levels, tmp_path, monkeypatch):
    """dcsg_export projects, downloads and writes large meshes as a pipeline over chunks of the triangle buffer (GPU on
    chunk k+1 while the host writes chunk k).  Projection is a pure function of a vertex's position, so the files and the
    device buffer must equal those of the one-after-the-other path bit for bit -- including with a different chunk count."""
    s = read_scene(name)
    assert ev.build(s["scene.txt"], s["buildprocedure.txt"], s["scene.cl"])[0] == 0
    cfg_txt = export_config_text(name, levels=list(levels), gd=50)
    monkeypatch.setenv("DCSG_NO_PIPELINE", "1")
    a = ev.export(cfg_txt, str(tmp_path / "a.stl"), str(tmp_path / "a.ply"))
    ta = ev.download_triangles()
    monkeypatch.delenv("DCSG_NO_PIPELINE")
    for chunks in (None, "7"):
        if chunks:
            monkeypatch.setenv("DCSG_EXPORT_CHUNKS", chunks)
        b = ev.export(cfg_txt, str(tmp_path / "b.stl"), str(tmp_path / "b.ply"))
        tb = ev.download_triangles()
        assert a.triangles == b.triangles and np.array_equal(ta.view(np.uint32), tb.view(np.uint32))
        assert (tmp_path / "a.stl").read_bytes() == (tmp_path / "b.stl").read_bytes()
        assert (tmp_path / "a.ply").read_bytes() == (tmp_path / "b.ply").read_bytes()
