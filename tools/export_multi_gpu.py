#!/usr/bin/env python3
"""Multi-GPU export (BASELINE config 4: Design2 1024^3 sharded over the GPUs of one node, stitched .ply).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/export_multi_gpu.py --scene-dir DIR [--levels MIN MAX GRID] --out PREFIX

One process per GPU; each rank loads the scene and runs the export on its Z slab.  Default: the node-wise sweep of the
slab's octree nodes (no exchange: every rank evaluates the plane it shares with its upper neighbour itself), projection,
all_gather of the triangle counts, and every rank writes its own byte range of PREFIX.stl / PREFIX.ply.
--dense-halo: the dense slab sweep with the halo plane over RCCL and the triangles gathered to rank 0 (round 1's path).
Prints one JSON line with counts and per-rank timings.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene-dir", required=True)
    ap.add_argument("--levels", type=int, nargs=3, default=None)
    ap.add_argument("--gd", type=int, default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--dense-normals", action="store_true")
    ap.add_argument("--check-halo", action="store_true")
    ap.add_argument("--dense-halo", action="store_true", help="dense slab sweep + halo plane over RCCL + gather to rank 0")
    ap.add_argument("--repeat", type=int, default=1, help="node-wise mode: best of this many exports")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal on a box with fewer GPUs than ranks: ranks share GPUs, planes and triangles "
                         "are staged through host memory")
    ap.add_argument("--verify-single", action="store_true",
                    help="rank 0 also exports the whole lattice on its own and checks that the stitched mesh is the "
                         "same triangle set")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from designcsg_amd.evaluator import Evaluator, parse_export_config, write_ply, write_stl
    from designcsg_amd.sharded import (HipBackend, HipSlab, read_ply_triangles, read_stl_triangles, sharded_export,
                                       sharded_export_files)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    lines = open(os.path.join(args.scene_dir, "exportConfig.txt")).read().split("\n")
    if args.levels:
        lines[1:4] = [str(v) for v in args.levels]
    if args.gd is not None:
        lines[5] = str(args.gd)
    cfg = parse_export_config("\n".join(lines))

    ev = Evaluator(local_rank)
    status, log = ev.build_from_dir(args.scene_dir)
    if status != 0:
        sys.exit("Error building export kernel:\n" + log)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    cfg_txt = "\n".join(lines)
    import numpy as np
    key = lambda t: np.sort(np.ascontiguousarray(t, dtype=np.float32).reshape(-1, 9).view(np.uint32).copy().view(
        [("", np.uint32)] * 9), axis=0)
    if args.dense_halo:
        t0 = time.perf_counter()
        res = sharded_export(HipBackend(ev, "cuda:%d" % local_rank), cfg, rank, world, dense_normals=args.dense_normals,
                             check_halo=args.check_halo, host_staged=args.backend == "gloo")
        t_compute = time.perf_counter() - t0
        if rank == 0:
            tris = res["triangles"]
            if args.out:
                write_stl(args.out + ".stl", tris)
                write_ply(args.out + ".ply", tris)
            out = {"n_gpus": world, "backend": args.backend, "mode": "dense slabs + halo plane + gather", "triangles": int(len(tris)),
                   "per_rank_triangles": res["counts"], "box": res["box"][:2], "export_s": round(time.perf_counter() - t0, 4),
                   "compute_and_gather_s": round(t_compute, 4)}
    else:
        stl = args.out + ".stl" if args.out else None
        ply = args.out + ".ply" if args.out else None
        best = None
        for _ in range(max(1, args.repeat)):
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            res = sharded_export_files(HipSlab(ev, dense_normals=args.dense_normals), cfg_txt, rank, world, stl, ply)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        st = res["stats"]
        mine = {"rank": rank, "triangles": res["counts"][rank], "swept_points": int(st.swept_points),
                "stage_ms": {k: round(getattr(st, "t_" + k) * 1e3, 2) for k in ("bbox", "sweep", "mesh", "project")}}
        if world > 1:
            every = [None] * world
            dist.all_gather_object(every, mine)
        else:
            every = [mine]
        if rank == 0:
            out = {"n_gpus": world, "backend": args.backend, "mode": "node-wise slabs, per-rank byte-range writes",
                   "triangles": res["total"], "per_rank_triangles": res["counts"], "box": [list(st.box_center), st.box_diameter],
                   "export_s": round(best, 4), "ranks": every}
            tris = None
            if args.out:
                out["stl_bytes"], out["ply_bytes"] = os.path.getsize(stl), os.path.getsize(ply)
                tris = read_stl_triangles(stl)
                pv, pf = read_ply_triangles(ply)
                # the two files hold the same triangles in the same order; faces are (3i, 3i+1, 3i+2)
                assert np.array_equal(pv.astype(np.float32).view(np.uint32), tris.view(np.uint32))
                assert np.array_equal(pf, np.arange(3 * len(tris), dtype=np.uint32).reshape(-1, 3))
                assert out["stl_bytes"] == 84 + 50 * res["total"] and out["ply_bytes"] == len(open(ply, "rb").read())
    if rank == 0:
        if args.verify_single:
            st1 = ev.export(cfg_txt)
            whole = ev.download_triangles()
            if tris is None:
                sys.exit("--verify-single needs --out (the files are what is verified) or --dense-halo")
            same = len(whole) == len(tris) and bool(np.array_equal(key(whole), key(tris)))
            out["equals_single_gpu_export_as_a_set"] = same
            out["single_gpu_triangles"] = int(st1.triangles)
            if not same:
                print(json.dumps(out))
                sys.exit("stitched mesh differs from the single-GPU mesh")
        print(json.dumps(out))
    ev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
