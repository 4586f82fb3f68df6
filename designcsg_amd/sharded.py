"""Z-slab sharded export: one process per GPU, torch.distributed (RCCL over xGMI on GPUs).

The lattice values are a pure function of position and octree subtrees are independent, so the export
shards by Z: rank r of N owns lattice planes [r*res/N, (r+1)*res/N) -- the last rank
also owns plane res -- sweeps them, and needs exactly ONE plane it does not own: plane (r+1)*res/N, the
upper face of its top cells, which its upper neighbour owns as its first plane.  That halo plane is the
only data-path exchange (point-to-point send/recv between Z neighbours); coarse octree nodes that
straddle slabs are processed redundantly by every rank from on-demand samples.  Triangles are then
gathered to rank 0 (counts by all_gather, payload by gather), which writes the files.

Two drivers share that partition.  `sharded_export_files` (default of tools/export_multi_gpu.py) is the export as the
single-GPU path runs it, per slab: node-wise sweep restricted to the slab's octree nodes (each rank evaluates the shared
plane itself: a sample is a pure function of its lattice index, so no exchange is needed), projection, then all_gather of
the triangle COUNTS only and every rank writes its own byte range of the .stl / .ply (fixed-size records).
`sharded_export` is the dense-slab form with the halo plane travelling over RCCL and the triangles gathered to rank 0.

The compute backend is an object with the four methods of `HipBackend`; the product always uses
`HipBackend` (libdcsg_hip through the C ABI).  tests/test_sharded_gloo.py injects a CPU checker backend to
run this driver's partition / halo / gather logic under gloo with world_size 2 -- there is no CPU
backend in this package and none is selected automatically.
"""
import numpy as np
import torch
import torch.distributed as dist


def slab_ranges(res, world):
    """[(lo, hi)] per rank: own planes lo..hi-1, halo plane hi; the last rank also owns plane res."""
    if world < 1 or res % world != 0:
        raise ValueError("grid resolution %d does not split into %d equal Z slabs" % (res, world))
    return [(r * res // world, (r + 1) * res // world) for r in range(world)]


class HipBackend:
    """The MI355X path: lattice slabs live in torch tensors on this rank's GPU, kernels run through
    libdcsg_hip on torch's current stream."""

    def __init__(self, evaluator, device):
        self.ev = evaluator
        self.device = torch.device(device)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def estimate_bounding_box(self, bb_half):
        return self.ev.estimate_bounding_box(bb_half)

    def sweep(self, center, diameter, res, z0, z1, sdf, normals):
        """Fill planes z0..z1-1 into the given slab tensors (normals may be None)."""
        self.ev.sweep_lattice_device(center, diameter, res, z0, z1, sdf.data_ptr(),
                                     0 if normals is None else normals.data_ptr(), self._stream())

    def alloc(self, n_floats):
        return torch.empty(n_floats, dtype=torch.float32, device=self.device)

    def mesh(self, center, diameter, cfg, sdf, normals, z_lo, z_hi, own_lo, own_hi):
        n, st = self.ev.mesh_lattice_device(center, diameter, cfg, sdf.data_ptr(),
                                            0 if normals is None else normals.data_ptr(), z_lo, z_hi, own_lo, own_hi,
                                            self._stream())
        return n, st

    def project_and_fetch(self, steps):
        """Projection of the meshed triangles, returned as a (T,3,3) float32 tensor on this device."""
        self.ev.project_triangles(steps, self._stream())
        _, n = self.ev.triangles_device()
        out = torch.empty((n, 3, 3), dtype=torch.float32, device=self.device)
        self.ev.copy_triangles_device(out.data_ptr(), n, self._stream())
        return out


def exchange_halo(sdf, normals, plane_floats, n_own, rank, world, group=None, host_staged=False):
    """Plane 0 of every rank > 0 goes to the halo plane (index n_own) of rank-1.  Point-to-point between Z
    neighbours: on xGMI each pair has a direct link, so nothing is routed through a ring.
    host_staged: the planes travel through host memory (gloo rehearsals of the N>1 logic, e.g. several ranks on
    one GPU); the product path (RCCL) sends the device tensors directly."""
    reqs, landing = [], []
    out = lambda t: t.cpu() if host_staged else t
    if rank > 0:
        reqs.append(dist.isend(out(sdf[:plane_floats]), rank - 1, group=group))
        if normals is not None:
            reqs.append(dist.isend(out(normals[:3 * plane_floats]), rank - 1, group=group))
    if rank < world - 1:
        dst = [sdf[n_own * plane_floats:(n_own + 1) * plane_floats]]
        if normals is not None:
            dst.append(normals[3 * n_own * plane_floats:3 * (n_own + 1) * plane_floats])
        for d in dst:
            buf = torch.empty(d.shape, dtype=d.dtype) if host_staged else d
            landing.append((d, buf))
            reqs.append(dist.irecv(buf, rank + 1, group=group))
    for r in reqs:
        r.wait()
    if host_staged:
        for d, buf in landing:
            d.copy_(buf)


def gather_triangles(tris, rank, world, group=None, host_staged=False):
    """Variable-length gather to rank 0: counts by all_gather, payload padded to the largest count.
    Returns (concatenated (T,3,3) tensor on rank 0 else None, list of per-rank counts)."""
    if host_staged:
        tris = tris.cpu()
    count = torch.tensor([tris.shape[0]], dtype=torch.int64, device=tris.device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    padded = torch.zeros((cap, 3, 3), dtype=torch.float32, device=tris.device)
    padded[:tris.shape[0]] = tris
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, bufs, dst=0, group=group)
    if rank != 0:
        return None, counts
    return torch.cat([b[:c] for b, c in zip(bufs, counts)]), counts


def sharded_export(backend, cfg, rank, world, dense_normals=False, group=None, check_halo=False, host_staged=False):
    """Runs the export stages on this rank's slab.  cfg: ExportConfig (parse_export_config).
    Returns dict(triangles=(T,3,3) CPU numpy array on rank 0 / None elsewhere, counts, box, stats)."""
    center, diameter, interior = backend.estimate_bounding_box(cfg.bounding_box_half_diameter)
    res = 1 << cfg.grid_level
    n1 = res + 1
    plane = n1 * n1
    lo, hi = slab_ranges(res, world)[rank]
    last = rank == world - 1
    n_own = hi - lo + (1 if last else 0)      # planes this rank computes
    n_res = hi - lo + 1                       # planes resident for meshing: lo..hi
    sdf = backend.alloc(n_res * plane)
    nrm = backend.alloc(3 * n_res * plane) if dense_normals else None
    # own planes (the halo plane hi is the neighbour's, except on the last rank)
    backend.sweep(center, diameter, res, lo, lo + n_own, sdf, nrm)
    if world > 1:
        exchange_halo(sdf, nrm, plane, hi - lo, rank, world, group, host_staged)
        if check_halo and not last:
            # self-check: the received plane must equal what this rank computes for it
            chk = backend.alloc(plane)
            backend.sweep(center, diameter, res, hi, hi + 1, chk, None)
            if not torch.equal(chk.view(torch.int32), sdf[(hi - lo) * plane:(hi - lo + 1) * plane].view(torch.int32)):
                raise RuntimeError("halo plane %d from rank %d differs from the local evaluation" % (hi, rank + 1))
    n_tris, stats = backend.mesh(center, diameter, cfg, sdf, nrm, lo, hi, lo, hi)
    tris = backend.project_and_fetch(cfg.gradient_descent_steps)
    assert tris.shape[0] == n_tris
    if world > 1:
        all_tris, counts = gather_triangles(tris, rank, world, group, host_staged)
    else:
        all_tris, counts = tris, [int(tris.shape[0])]
    return dict(triangles=None if all_tris is None else all_tris.cpu().numpy(), counts=counts,
                box=(center, diameter, interior), stats=stats, slab=(lo, hi))


class HipSlab:
    """The MI355X path of sharded_export_files: libdcsg_hip's slab entry points on this rank's GPU."""

    def __init__(self, evaluator, dense_normals=False, dense_sweep=False):
        self.ev, self.kw = evaluator, dict(dense_normals=dense_normals, dense_sweep=dense_sweep)

    def export_slab(self, cfg_txt, rank, world):
        st = self.ev.export_slab(cfg_txt, rank, world, **self.kw)
        return int(st.triangles), st

    def write_slab(self, stl_path, ply_path, first, total):
        self.ev.write_slab(stl_path, ply_path, first, total, 0)


def sharded_export_files(slab, cfg_txt, rank, world, stl_path, ply_path, group=None):
    """Every rank meshes and projects its Z slab, then writes its own byte range of the files.  The only collectives are
    the all_gather of the per-rank triangle counts and two barriers around the payload writes (rank 0 opens the files and
    marks them incomplete before, and gives them their final length and header after).  All ranks must see the same
    file system (one node).  Returns dict(counts, first, total, stats)."""
    from .evaluator import WRITE_BEGIN, WRITE_END, write_range
    n, st = slab.export_slab(cfg_txt, rank, world)
    if world > 1:
        mine = torch.tensor([n], dtype=torch.int64)
        # counts travel on the host: 8 bytes per rank, and gloo / RCCL both take a CPU or a device tensor of their own kind
        backend = dist.get_backend(group)
        if backend == "nccl":
            mine = mine.cuda()
        counts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(counts, mine, group=group)
        counts = [int(c.item()) for c in counts]
    else:
        counts = [n]
    first, total = sum(counts[:rank]), sum(counts)
    paths = [p for p in (stl_path, ply_path) if p is not None]
    if rank == 0:
        for p in paths:
            write_range(p, None, 0, total, WRITE_BEGIN)
    if world > 1:
        dist.barrier(group=group)
    slab.write_slab(stl_path, ply_path, first, total)
    if world > 1:
        dist.barrier(group=group)
    if rank == 0:
        for p in paths:
            write_range(p, None, 0, total, WRITE_END)
    return dict(counts=counts, first=first, total=total, stats=st)


def read_stl_triangles(path):
    """Triangles of a binary STL written by the export ((x, z, y) on disk, utils.hpp:41-103) as a (T,3,3) float32 array."""
    raw = np.fromfile(path, dtype=np.uint8)
    n = int(raw[80:84].view("<u4")[0])
    rec = raw[84:84 + 50 * n].reshape(n, 50)[:, 12:48].copy().view("<f4").reshape(n, 3, 3)
    return np.ascontiguousarray(rec[:, :, [0, 2, 1]])


def read_ply_triangles(path):
    """Triangles of the reference-layout PLY (un-indexed doubles + uchar/uint faces, utils.hpp:106-154): (T,3,3) float64
    and the (T,3) face indices."""
    raw = open(path, "rb").read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    head = raw[:end].decode()
    nv = int(head.split("element vertex ")[1].split("\n")[0])
    nf = int(head.split("element face ")[1].split("\n")[0])
    v = np.frombuffer(raw, dtype="<f8", count=3 * nv, offset=end).reshape(-1, 3, 3)
    f = np.frombuffer(raw, dtype=np.uint8, count=13 * nf, offset=end + 24 * nv).reshape(nf, 13)
    assert len(raw) == end + 24 * nv + 13 * nf and (f[:, 0] == 3).all()
    return v, f[:, 1:].copy().view("<u4").reshape(nf, 3)
