"""N > 1 path on CPU: world_size-2 gloo run of the Z-slab driver (partition, halo exchange, gather)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, scene_dir


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_slab_ranges():
    from designcsg_amd.sharded import slab_ranges
    assert slab_ranges(512, 1) == [(0, 512)]
    assert slab_ranges(512, 8) == [(64 * r, 64 * r + 64) for r in range(8)]
    with pytest.raises(ValueError):
        slab_ranges(512, 3)


@pytest.mark.parametrize("dense", ["0", "1"])
def test_two_rank_gloo_export_equals_single(oracle, lookup_table, tmp_path, dense):
    env = dict(os.environ, DENSE=dense, OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "sharded_gloo_worker.py"), str(tmp_path), scene_dir("sphere")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    info = [json.load(open(tmp_path / ("r%d.json" % k))) for k in range(2)]
    assert [tuple(i["slab"]) for i in info] == [(0, 64), (64, 128)]
    assert info[0]["counts"] == info[1]["counts"] and sum(info[0]["counts"]) == 33608  # SURVEY 8(c), sphere 4/6/7
    got = np.load(tmp_path / "tris.npy")
    # single-process result from the oracle
    sc = oracle.Scene.from_dir(scene_dir("sphere"))
    cfg = oracle.parse_export_config("10.0\n4\n6\n7\n0.7853981633974483\n5\n16\n512\n4096\n")
    c, d, _ = sc.estimate_bounding_box(cfg.bounding_box_half_diameter)
    raw, _ = sc.get_surface(c, d[0], cfg, lookup_table)
    want = sc.gradient_descent(raw, 5)
    key = lambda t: np.sort(np.ascontiguousarray(t, dtype=np.float32).reshape(-1, 9).view(np.uint32).copy()
                            .view([("", np.uint32)] * 9), axis=0)
    assert np.array_equal(key(got), key(want))
    # rank order is preserved: rank 0's triangles (lower slab) come first
    n0 = info[0]["counts"][0]
    assert got[:n0, :, 2].mean() < got[n0:, :, 2].mean()


def test_two_rank_gloo_byte_range_files(oracle, lookup_table, built_lib, tmp_path):
    """sharded_export_files under gloo, world_size 2: counts by all_gather, rank 0 marks the files incomplete, every rank
    writes its byte range through the product's range writers, rank 0 finishes them.  The files must be what the
    single-process writers produce for the same triangles in rank order."""
    from designcsg_amd.evaluator import write_ply, write_stl
    from designcsg_amd.sharded import read_ply_triangles, read_stl_triangles
    env = dict(os.environ, MODE="files", OMP_NUM_THREADS="4")
    # stale, longer files of the same names: the tails must go
    (tmp_path / "o.stl").write_bytes(b"x" * 5_000_000)
    (tmp_path / "o.ply").write_bytes(b"ply\n" + b"y" * 9_000_000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "sharded_gloo_worker.py"), str(tmp_path), scene_dir("sphere")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    info = [json.load(open(tmp_path / ("r%d.json" % k))) for k in range(2)]
    counts = info[0]["counts"]
    assert info[1]["counts"] == counts and sum(counts) == 33608 == info[0]["total"]  # SURVEY 8(c), sphere 4/6/7
    assert info[0]["first"] == 0 and info[1]["first"] == counts[0]
    tris = read_stl_triangles(tmp_path / "o.stl")
    assert len(tris) == 33608 and os.path.getsize(tmp_path / "o.stl") == 84 + 50 * 33608
    # same bytes as the one-shot writers on the same triangles in the same order
    write_stl(tmp_path / "w.stl", tris)
    write_ply(tmp_path / "w.ply", tris)
    assert (tmp_path / "w.stl").read_bytes() == (tmp_path / "o.stl").read_bytes()
    assert (tmp_path / "w.ply").read_bytes() == (tmp_path / "o.ply").read_bytes()
    pv, pf = read_ply_triangles(tmp_path / "o.ply")
    assert np.array_equal(pf, np.arange(3 * 33608, dtype=np.uint32).reshape(-1, 3))
    # the triangle set is the single-process export's
    sc = oracle.Scene.from_dir(scene_dir("sphere"))
    cfg = oracle.parse_export_config("10.0\n4\n6\n7\n0.7853981633974483\n5\n16\n512\n4096\n")
    c, d, _ = sc.estimate_bounding_box(cfg.bounding_box_half_diameter)
    raw, _ = sc.get_surface(c, d[0], cfg, lookup_table)
    want = sc.gradient_descent(raw, 5)
    key = lambda t: np.sort(np.ascontiguousarray(t, dtype=np.float32).reshape(-1, 9).view(np.uint32).copy()
                            .view([("", np.uint32)] * 9), axis=0)
    assert np.array_equal(key(tris), key(want))
    assert tris[:counts[0], :, 2].mean() < tris[counts[0]:, :, 2].mean()   # rank order = slab order
