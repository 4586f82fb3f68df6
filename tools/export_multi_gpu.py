#!/usr/bin/env python3
"""Multi-GPU export (BASELINE config 4: Design2 1024^3 sharded over the GPUs of one node, stitched .ply).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/export_multi_gpu.py --scene-dir DIR [--levels MIN MAX GRID] --out PREFIX

One process per GPU; each rank loads the scene, sweeps and meshes its Z slab (one halo plane from its upper
neighbour over RCCL), projects its vertices; rank 0 gathers the triangles in slab order and writes
PREFIX.stl / PREFIX.ply.  Prints one JSON line with counts and per-rank timings.
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
    from designcsg_amd.sharded import HipBackend, sharded_export

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
    t0 = time.perf_counter()
    res = sharded_export(HipBackend(ev, "cuda:%d" % local_rank), cfg, rank, world, dense_normals=args.dense_normals,
                         check_halo=args.check_halo, host_staged=args.backend == "gloo")
    t_compute = time.perf_counter() - t0
    if rank == 0:
        tris = res["triangles"]
        if args.out:
            write_stl(args.out + ".stl", tris)
            write_ply(args.out + ".ply", tris)
        out = {"n_gpus": world, "backend": args.backend, "triangles": int(len(tris)), "per_rank_triangles": res["counts"],
               "box": res["box"][:2], "export_s": round(time.perf_counter() - t0, 4),
               "compute_and_gather_s": round(t_compute, 4)}
        if args.verify_single:
            import numpy as np
            whole = sharded_export(HipBackend(ev, "cuda:%d" % local_rank), cfg, 0, 1, dense_normals=args.dense_normals)
            key = lambda t: np.sort(np.ascontiguousarray(t, dtype=np.float32).reshape(-1, 9).view(np.uint32).copy().view(
                [("", np.uint32)] * 9), axis=0)
            same = len(whole["triangles"]) == len(tris) and bool(np.array_equal(key(whole["triangles"]), key(tris)))
            out["equals_single_gpu_export_as_a_set"] = same
            if not same:
                print(json.dumps(out))
                sys.exit("stitched mesh differs from the single-GPU mesh")
        print(json.dumps(out))
    ev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
