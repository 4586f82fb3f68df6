"""Host-side mirror of the reference's evaluator / export interface over libdcsg_hip.so.

`Evaluator` keeps the names, argument meaning and error behaviour of the reference class
(Evaluator.h:20-57, Evaluator.cpp) so that parity tests read like calls into the original:

    ev = Evaluator(device=0)
    status, log = ev.build(scene_txt, buildprocedure_txt, scene_cl)   # (0, "Success!") / (-1, build log)
    ev.setArbitraryData(data, items)
    sdf = ev.eval_sdf_at_points(points)          # n floats
    nrm = ev.eval_normal_at_points(points)       # n x 3 floats

and adds the export stages the reference runs on the host around it (DesignCSG.cpp:638-790).
Everything computes on the GPU through the C ABI; numpy is only the container.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import DcsgError, ExportConfig, ExportStats, MeshStats


def _fptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f3(v):
    return (C.c_float * 3)(float(v[0]), float(v[1]), float(v[2]))


def read_arbitrary_data(path):
    """arbitrary_data.hex: 131072 little-endian f32 (updateExportArbitraryData, DesignCSG.cpp:507-529)."""
    return np.fromfile(path, dtype="<f4")


class Evaluator:
    def __init__(self, device=0):
        self._L = _lib.lib()
        h = C.c_void_p()
        rc = self._L.dcsg_create(C.byref(h), int(device))
        if rc != 0:
            raise DcsgError(rc, self._L.dcsg_last_error(None).decode(errors="replace"))
        self._h = h
        self.build_success = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.dcsg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        return _lib.check(self._h, rc)

    def set_numerics(self, mode):
        """Arithmetic of the scenes built after this call: "exact" (default; bit-identical to the reference arithmetic)
        or "relaxed" (opt-in: hardware sqrt / rsq, FMA contraction; results within ~1e-6)."""
        self._ck(self._L.dcsg_set_numerics(self._h, {"exact": 0, "relaxed": 1}[mode]))

    # ---- Evaluator::build (Evaluator.cpp:45-112) -------------------------------------------
    def build(self, scene_txt, buildprocedure_txt, scene_cl=None, arbitrary_data=None):
        ad = None
        n_ad = 0
        if arbitrary_data is not None:
            ad = np.ascontiguousarray(arbitrary_data, dtype=np.float32)
            n_ad = ad.size
        rc = self._L.dcsg_load_scene(self._h, scene_txt.encode(), buildprocedure_txt.encode(),
                                     None if scene_cl is None else scene_cl.encode(),
                                     None if ad is None else _fptr(ad), n_ad)
        if rc == _lib.ERR_BUILD:
            self.build_success = 0
            return -1, self._L.dcsg_last_error(self._h).decode(errors="replace")
        self._ck(rc)
        self.build_success = 1
        return 0, "Success!"

    def build_from_dir(self, scene_dir, use_scene_cl=True):
        """Load the files a design run leaves in its working directory."""
        rd = lambda n: open(os.path.join(scene_dir, n)).read()
        scl = rd("scene.cl") if use_scene_cl and os.path.exists(os.path.join(scene_dir, "scene.cl")) else None
        ad = None
        for name in ("arbitrary_data.hex", "arbitrary_data.prefix.hex"):  # the latter: fixture with the zero tail cut
            if os.path.exists(os.path.join(scene_dir, name)):
                ad = read_arbitrary_data(os.path.join(scene_dir, name))
                break
        return self.build(rd("scene.txt"), rd("buildprocedure.txt"), scl, ad)

    def load_profile(self):
        """Host-side milliseconds of dcsg_create and of the last dcsg_load_scene, by stage."""
        parse = lambda b: {k: float(v) for k, v in (e.split("=") for e in b.decode().split(";") if e)}
        return {"create": parse(self._L.dcsg_load_profile(self._h, 0)), "load_scene": parse(self._L.dcsg_load_profile(self._h, 1))}

    def scene_info(self):
        v = [C.c_int() for _ in range(6)]
        self._ck(self._L.dcsg_scene_info(self._h, *[C.byref(x) for x in v]))
        return dict(objects=v[0].value, steps=v[1].value, slots=v[2].value, jit=bool(v[3].value),
                    interpreter=v[4].value, stack_depth=v[5].value)

    def scene_flags(self):
        """Diagnostic: flags word of the scene's device tables (bit 1: the root is a plain union, dcsg.h)."""
        f = C.c_int()
        self._ck(self._L.dcsg_debug_scene_flags(self._h, C.byref(f)))
        return f.value

    # ---- Evaluator::setArbitraryData (Evaluator.cpp:213-225) ------------------------------
    def setArbitraryData(self, data, items=None):
        a = np.ascontiguousarray(data, dtype=np.float32)
        self._ck(self._L.dcsg_set_arbitrary_data(self._h, _fptr(a), a.size if items is None else int(items)))

    # ---- Evaluator::eval_*_at_points (Evaluator.cpp:117-211) -------------------------------
    def eval_sdf_at_points(self, points):
        p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        out = np.empty(len(p), dtype=np.float32)
        self._ck(self._L.dcsg_eval_sdf(self._h, _fptr(p), len(p), _fptr(out)))
        return out

    def eval_normal_at_points(self, points):
        p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        out = np.empty((len(p), 3), dtype=np.float32)
        self._ck(self._L.dcsg_eval_normal(self._h, _fptr(p), len(p), _fptr(out)))
        return out

    # ---- dense lattice (ISV.hpp) -------------------------------------------------------------
    def eval_lattice(self, center, diameter, res, z0=0, z1=None, sdf=True, normals=True):
        n1 = res + 1
        z1 = n1 if z1 is None else z1
        d = [diameter] * 3 if np.isscalar(diameter) else diameter
        s = np.empty((z1 - z0, n1, n1), dtype=np.float32) if sdf else None
        n = np.empty((z1 - z0, n1, n1, 3), dtype=np.float32) if normals else None
        self._ck(self._L.dcsg_eval_lattice(self._h, _f3(center), _f3(d), res, z0, z1,
                                           _fptr(s) if sdf else None, _fptr(n) if normals else None))
        return s, n

    def sweep_lattice_device(self, center, diameter, res, z0, z1, d_sdf, d_normals, stream=0):
        """Device-pointer sweep (torch tensors' data_ptr()); asynchronous on `stream`."""
        d = [diameter] * 3 if np.isscalar(diameter) else diameter
        self._ck(self._L.dcsg_sweep_lattice_device(self._h, _f3(center), _f3(d), res, z0, z1,
                                                   C.c_void_p(d_sdf or None), C.c_void_p(d_normals or None),
                                                   C.c_void_p(stream or None)))

    def set_sweep_timing(self, on=True):
        """Timing events around the sweeps of sweep_lattice_device (off by default; last_sweep_ms needs them)."""
        self._ck(self._L.dcsg_set_sweep_timing(self._h, 1 if on else 0))

    def last_sweep_ms(self):
        ms = C.c_double()
        self._ck(self._L.dcsg_last_sweep_ms(self._h, C.byref(ms)))
        return ms.value

    # ---- export stages (DesignCSG.cpp:638-790) ------------------------------------------------
    def estimate_bounding_box(self, bounding_box_half_diameter):
        c = (C.c_float * 3)()
        d = C.c_float()
        n = C.c_uint64()
        self._ck(self._L.dcsg_estimate_bounding_box(self._h, bounding_box_half_diameter, c, C.byref(d), C.byref(n)))
        return [c[0], c[1], c[2]], d.value, n.value

    def set_lookup_table(self, text):
        self._ck(self._L.dcsg_set_lookup_table(self._h, text.encode()))

    def mesh_lattice_device(self, center, diameter, cfg, d_sdf, d_normals, z_lo, z_hi, z_own_lo, z_own_hi,
                            stream=0):
        n = C.c_uint64()
        st = MeshStats()
        self._ck(self._L.dcsg_mesh_lattice_device(self._h, _f3(center), diameter, C.byref(cfg),
                                                  C.c_void_p(d_sdf or None), C.c_void_p(d_normals or None),
                                                  z_lo, z_hi, z_own_lo, z_own_hi, C.byref(n), C.byref(st),
                                                  C.c_void_p(stream or None)))
        return n.value, st

    def project_triangles(self, steps, stream=0):
        self._ck(self._L.dcsg_project_triangles(self._h, steps, C.c_void_p(stream or None)))

    def project_vertices_device(self, d_xyz, n_vertices, steps, stream=0):
        self._ck(self._L.dcsg_project_vertices_device(self._h, C.c_void_p(d_xyz), n_vertices, steps,
                                                      C.c_void_p(stream or None)))

    def triangles_device(self):
        p = C.c_void_p()
        n = C.c_uint64()
        self._ck(self._L.dcsg_triangles_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def download_triangles(self):
        _, n = self.triangles_device()
        out = np.empty((n, 3, 3), dtype=np.float32)
        self._ck(self._L.dcsg_download_triangles(self._h, _fptr(out), n))
        return out

    def copy_triangles_device(self, d_out, max_triangles, stream=0):
        self._ck(self._L.dcsg_copy_triangles_device(self._h, C.c_void_p(d_out), max_triangles,
                                                    C.c_void_p(stream or None)))

    def debug_cells(self):
        """Diagnostic: cell lists of the loaded many-object scene (grid, cells built, entries handed out, build ms so far)."""
        g, nb, e, ms = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_double()
        self._ck(self._L.dcsg_debug_cells(self._h, C.byref(g), C.byref(nb), C.byref(e), C.byref(ms)))
        return dict(grid=g.value, cells_built=nb.value, entries=e.value, build_ms=round(ms.value, 3))

    def debug_sqrt_device(self, d_in, n, d_out, stream=0):
        self._ck(self._L.dcsg_debug_sqrt_device(self._h, C.c_void_p(d_in), n, C.c_void_p(d_out), C.c_void_p(stream or None)))

    def debug_normal_device(self, d_in, n, dy, dz, mixed, d_out, stream=0):
        """Diagnostic: normals of the gradients (in[i], dy, dz) as the kernels' get_normal tail computes them."""
        self._ck(self._L.dcsg_debug_normal_device(self._h, C.c_void_p(d_in), n, dy, dz, int(mixed), C.c_void_p(d_out),
                                                  C.c_void_p(stream or None)))

    def render_preview(self, campos=(0.0, 0.0, -15.0), right=(1.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), forward=(0.0, 0.0, 1.0)):
        """Kernel k1: the 640x480 preview of the loaded scene as a (480, 640, 3) uint8 array."""
        v = [np.ascontiguousarray(a, dtype=np.float32) for a in (campos, right, up, forward)]
        out = np.zeros((480, 640, 3), dtype=np.uint8)
        self._ck(self._L.dcsg_render_preview(self._h, _fptr(v[0]), _fptr(v[1]), _fptr(v[2]), _fptr(v[3]),
                                             out.ctypes.data_as(C.c_void_p)))
        return out

    def retopologize(self, center, diameter, min_level, grid_level, stream=0):
        """Opt-in: cms::retopologize as intended (T-junction vertices join the polygons of the triangles whose edges
        they lie on) on the context's triangles.  Returns the new triangle count."""
        self._ck(self._L.dcsg_retopologize(self._h, _f3(center), diameter, min_level, grid_level, C.c_void_p(stream or None)))
        return self.triangles_device()[1]

    def upload_triangles(self, tris):
        t = np.ascontiguousarray(tris, dtype=np.float32)
        self._ck(self._L.dcsg_upload_triangles(self._h, _fptr(t) if t.size else None, t.size // 9))

    def export(self, export_config_txt, stl_path=None, ply_path=None, dense_normals=False, indexed_ply=False,
               stl_normals=False, dense_sweep=False, retopologize=False):
        """MyFrame::OnExportInner end to end on this GPU.  indexed_ply / stl_normals select the opt-in writers
        (shared vertices; real facet normals) instead of the reference's file layouts."""
        st = ExportStats()
        flags = (_lib.EXPORT_DENSE_NORMALS if dense_normals else 0) | (_lib.EXPORT_INDEXED_PLY if indexed_ply else 0) | \
            (_lib.EXPORT_STL_NORMALS if stl_normals else 0) | (_lib.EXPORT_DENSE_SWEEP if dense_sweep else 0) | \
            (_lib.EXPORT_RETOPO if retopologize else 0)
        self._ck(self._L.dcsg_export(self._h, export_config_txt.encode(),
                                     None if stl_path is None else os.fsencode(stl_path),
                                     None if ply_path is None else os.fsencode(ply_path), flags, C.byref(st)))
        return st


    def export_slab(self, export_config_txt, rank, world, dense_normals=False, dense_sweep=False, equal_slabs=False):
        """Z slab `rank` of `world` of the export (work-balanced cut planes unless equal_slabs), stages A .. projection;
        triangles stay on the device.  st.slab_z_lo / st.slab_z_hi: the lattice planes owned."""
        st = ExportStats()
        flags = (_lib.EXPORT_DENSE_NORMALS if dense_normals else 0) | (_lib.EXPORT_DENSE_SWEEP if dense_sweep else 0) | \
            (_lib.EXPORT_EQUAL_SLABS if equal_slabs else 0)
        self._ck(self._L.dcsg_export_slab(self._h, export_config_txt.encode(), int(rank), int(world), flags, C.byref(st)))
        return st

    def write_slab(self, stl_path, ply_path, first, total, flags=0):
        """This context's triangles as triangles [first, first + n) of files holding `total` (range writers)."""
        self._ck(self._L.dcsg_write_slab(self._h, None if stl_path is None else os.fsencode(stl_path),
                                         None if ply_path is None else os.fsencode(ply_path), int(first), int(total), int(flags)))


def export_multi(evaluators, export_config_txt, stl_path=None, ply_path=None, dense_normals=False):
    """dcsg_export_multi: the sharded export in ONE process, evaluators[i] (one per GPU, same scene loaded) = Z slab i.
    Returns the per-slab ExportStats."""
    n = len(evaluators)
    hs = (C.c_void_p * n)(*[e._h for e in evaluators])
    st = (ExportStats * n)()
    rc = _lib.lib().dcsg_export_multi(hs, n, export_config_txt.encode(), None if stl_path is None else os.fsencode(stl_path),
                                      None if ply_path is None else os.fsencode(ply_path),
                                      _lib.EXPORT_DENSE_NORMALS if dense_normals else 0, st)
    _lib.check(evaluators[0]._h, rc)
    return list(st)


WRITE_BEGIN, WRITE_END = 1, 2


def write_range(path, tris, first, total, flags=0):
    """dcsg_write_stl_range / dcsg_write_ply_range by file extension; tris may be None (BEGIN / END only)."""
    fn = _lib.lib().dcsg_write_ply_range if str(path).endswith(".ply") else _lib.lib().dcsg_write_stl_range
    t = None if tris is None else np.ascontiguousarray(tris, dtype=np.float32)
    rc = fn(os.fsencode(path), None if t is None or t.size == 0 else _fptr(t), 0 if t is None else t.size // 9, int(first),
            int(total), int(flags))
    if rc:
        raise DcsgError(rc, "cannot write " + str(path))


def parse_export_config(text):
    cfg = ExportConfig()
    if _lib.lib().dcsg_parse_export_config(text.encode(), C.byref(cfg)) != 0:
        raise ValueError("malformed exportConfig.txt")
    return cfg


def write_stl(path, tris):
    t = np.ascontiguousarray(tris, dtype=np.float32)
    rc = _lib.lib().dcsg_write_stl(os.fsencode(path), _fptr(t), t.size // 9)
    if rc:
        raise DcsgError(rc, "cannot write " + str(path))


def write_ply(path, tris):
    t = np.ascontiguousarray(tris, dtype=np.float32)
    rc = _lib.lib().dcsg_write_ply(os.fsencode(path), _fptr(t), t.size // 9)
    if rc:
        raise DcsgError(rc, "cannot write " + str(path))


def write_stl_normals(path, tris):
    """Opt-in: STL with facet normals (the reference writes zero normals)."""
    t = np.ascontiguousarray(tris, dtype=np.float32)
    rc = _lib.lib().dcsg_write_stl_normals(os.fsencode(path), _fptr(t), t.size // 9)
    if rc:
        raise DcsgError(rc, "cannot write " + str(path))


def write_ply_indexed(path, tris):
    """Opt-in: PLY with shared vertices (the reference repeats every vertex per face).  Returns the vertex count."""
    t = np.ascontiguousarray(tris, dtype=np.float32)
    nv = C.c_uint64(0)
    rc = _lib.lib().dcsg_write_ply_indexed(os.fsencode(path), _fptr(t), t.size // 9, C.byref(nv))
    if rc:
        raise DcsgError(rc, "cannot write " + str(path))
    return nv.value
