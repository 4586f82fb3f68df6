"""Worker of tests/test_sharded_gloo.py: runs designcsg_amd.sharded.sharded_export under gloo on CPU with a
CHECKER backend built on the oracle (test infrastructure).  The backend verifies, inside mesh(), that the
slab this rank holds -- its own planes plus the halo plane received from its neighbour -- is bit-identical
to the oracle's dense lattice, then hands back the oracle's triangles that fall into the rank's Z slab."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from designcsg_amd import sharded  # noqa: E402
from oracle import oracle as O  # noqa: E402


class OracleCheckerBackend:
    def __init__(self, scene_dir, table):
        self.scene = O.Scene.from_dir(scene_dir)
        self.table = table
        self.halo_checked = False

    def estimate_bounding_box(self, bb_half):
        c, d, n = self.scene.estimate_bounding_box(bb_half)
        return c, d[0], n

    def alloc(self, n_floats):
        return torch.full((n_floats,), float("nan"), dtype=torch.float32)

    def sweep(self, center, diameter, res, z0, z1, sdf, normals):
        s, n = self.scene.eval_lattice(center, diameter, res, z0, z1, sdf=True, normals=normals is not None)
        # contract of HipBackend.sweep: plane z0 lands at the start of the tensors that were passed
        sdf[:s.size] = torch.from_numpy(s.reshape(-1))
        if normals is not None:
            normals[:n.size] = torch.from_numpy(n.reshape(-1))

    def mesh(self, center, diameter, cfg, sdf, normals, z_lo, z_hi, own_lo, own_hi):
        res = 1 << cfg.grid_level
        want, want_n = self.scene.eval_lattice(center, diameter, res, z_lo, z_hi + 1, sdf=True,
                                               normals=normals is not None)
        assert np.array_equal(sdf.numpy().view(np.uint32), want.reshape(-1).view(np.uint32)), \
            "resident slab (own planes + halo) differs from the dense lattice"
        if normals is not None:
            assert np.array_equal(normals.numpy().view(np.uint32), want_n.reshape(-1).view(np.uint32))
        self.halo_checked = True
        tris, st = self.scene.get_surface(center, diameter, cfg, self.table)
        z = tris[:, :, 2].astype(np.float64).mean(axis=1)
        z_of = lambda i: float(O.lattice_point(center, diameter, res, 0, 0, i)[2])
        keep = (z >= z_of(own_lo)) & (z < z_of(own_hi))
        self._tris = tris[keep]
        return int(keep.sum()), st

    def project_and_fetch(self, steps):
        return torch.from_numpy(self.scene.gradient_descent(self._tris, steps))


class OracleCheckerSlab:
    """Checker for sharded_export_files: the slab's triangles come from the oracle (those whose centroid lies in the
    rank's Z range), the files are written by the PRODUCT's range writers (host code of libdcsg_hip, no GPU needed) --
    the offsets, the BEGIN / END protocol and the barriers of the driver are what this exercises."""

    def __init__(self, scene_dir, table):
        self.scene, self.table, self.tris = O.Scene.from_dir(scene_dir), table, None

    def export_slab(self, cfg_txt, rank, world):
        cfg = O.parse_export_config(cfg_txt)
        c, d, _ = self.scene.estimate_bounding_box(cfg.bounding_box_half_diameter)
        res = 1 << cfg.grid_level
        tris, st = self.scene.get_surface(c, d[0], cfg, self.table)
        z = tris[:, :, 2].astype(np.float64).mean(axis=1)
        z_of = lambda i: float(O.lattice_point(c, d[0], res, 0, 0, i)[2])
        lo, hi = rank * res // world, (rank + 1) * res // world
        keep = (z >= z_of(lo)) & ((z < z_of(hi)) | (rank == world - 1))
        self.tris = self.scene.gradient_descent(tris[keep], cfg.gradient_descent_steps)
        return len(self.tris), st

    def write_slab(self, stl_path, ply_path, first, total):
        from designcsg_amd.evaluator import write_range
        for path in (stl_path, ply_path):
            if path is not None:
                write_range(path, self.tris, first, total, 0)


def main():
    out_dir, scene_dir = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    table = O.LookupTable(open(os.path.join(ROOT, "designcsg_amd", "data", "lookupTable.txt")).read())
    cfg_txt = "10.0\n4\n6\n7\n0.7853981633974483\n5\n16\n512\n4096\n"
    if os.environ.get("MODE") == "files":
        res = sharded.sharded_export_files(OracleCheckerSlab(scene_dir, table), cfg_txt, rank, world,
                                           os.path.join(out_dir, "o.stl"), os.path.join(out_dir, "o.ply"))
        json.dump({"rank": rank, "counts": res["counts"], "first": res["first"], "total": res["total"]},
                  open(os.path.join(out_dir, "r%d.json" % rank), "w"))
        dist.destroy_process_group()
        return
    from oracle.oracle import parse_export_config
    cfg = parse_export_config(cfg_txt)
    backend = OracleCheckerBackend(scene_dir, table)
    res = sharded.sharded_export(backend, cfg, rank, world, dense_normals=(os.environ.get("DENSE") == "1"),
                                 check_halo=True, host_staged=(os.environ.get("DENSE") == "1"))  # both transports
    assert backend.halo_checked
    json.dump({"rank": rank, "slab": res["slab"], "counts": res["counts"]}, open(os.path.join(out_dir, "r%d.json" % rank), "w"))
    if rank == 0:
        np.save(os.path.join(out_dir, "tris.npy"), res["triangles"])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
