"""ctypes binding of the CPU oracle (oracle/dcsg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by designcsg_amd.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(BUILD, "libdcsg_oracle.so")

BANK_FN = C.CFUNCTYPE(C.c_float, C.c_float, C.c_float, C.c_float, C.c_int)


class Box(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("diameters", C.c_float * 3)]

    @staticmethod
    def make(center, diameter):
        b = Box()
        d = [diameter] * 3 if np.isscalar(diameter) else list(diameter)
        for k in range(3):
            b.center[k] = float(center[k])
            b.diameters[k] = float(d[k])
        return b


class ExportConfig(C.Structure):
    _fields_ = [("bounding_box_half_diameter", C.c_float),
                ("min_level", C.c_int), ("max_level", C.c_int), ("grid_level", C.c_int),
                ("complex_surface_threshold", C.c_float),
                ("gradient_descent_steps", C.c_int),
                ("cache_subdivision", C.c_int), ("queries_before_gc", C.c_int),
                ("queries_before_free", C.c_int)]


class MeshStats(C.Structure):
    _fields_ = [("nodes_visited", C.c_uint64 * 32), ("nodes_culled", C.c_uint64 * 32),
                ("nodes_subdivided", C.c_uint64 * 32), ("triangles", C.c_uint64 * 32),
                ("sdf_samples", C.c_uint64), ("normal_samples", C.c_uint64)]


def build(force=False):
    """Compile the C oracle (gcc).  Building the checker is not using it."""
    src = [os.path.join(HERE, f) for f in ("dcsg_oracle.c", "dcsg_oracle.h")]
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= max(os.path.getmtime(s) for s in src)):
        return LIB
    subprocess.run(["make", "-C", HERE, "-s"], check=True)
    return LIB


REFERENCE = "/root/reference/master"
REF_LIB = os.path.join(HERE, "_ref", "libdcsg_ref.so")


def build_ref(force=False):
    """oracle/_ref: the parts of the REAL reference that compile on their own (ref_shim.cpp), where the reference
    tree is present (the build container; the built file travels to the GPU box).  Returns the path or None."""
    if not os.path.isdir(REFERENCE):
        return REF_LIB if os.path.exists(REF_LIB) else None
    shims = [os.path.join(HERE, f) for f in ("ref_shim.cpp", "ref_isv_shim.cpp", "Makefile")]
    if force or not os.path.exists(REF_LIB) or os.path.getmtime(REF_LIB) < max(os.path.getmtime(f) for f in shims):
        subprocess.run(["make", "-C", HERE, "-s", "ref"], check=True)
    return REF_LIB


def scene_bank_path(scene_cl_text):
    h = hashlib.sha256(scene_cl_text.encode()).hexdigest()[:16]
    return os.path.join(BUILD, "scene_%s.so" % h)


def build_scene_bank(scene_cl_path):
    """Compile one generated scene.cl to a host shared object (clang -x cl)."""
    text = open(scene_cl_path).read()
    out = scene_bank_path(text)
    if not os.path.exists(out):
        os.makedirs(BUILD, exist_ok=True)
        subprocess.run(["make", "-C", HERE, "-s", "scene", "SCENE_CL=" + os.path.abspath(scene_cl_path),
                        "OUT=" + out], check=True)
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.oracle_scene_parse.restype = C.c_void_p
        L.oracle_scene_parse.argtypes = [C.c_char_p, C.c_char_p]
        L.oracle_scene_free.argtypes = [C.c_void_p]
        L.oracle_scene_set_bank.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_builtin_bank.restype = C.c_float
        L.oracle_builtin_bank.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int]
        L.oracle_parse_export_config.argtypes = [C.c_char_p, C.POINTER(ExportConfig)]
        L.oracle_primary_sdf.restype = C.c_float
        L.oracle_primary_sdf.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.oracle_get_normal.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.oracle_eval_sdf.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.oracle_eval_normal.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.oracle_lattice_point.argtypes = [C.POINTER(Box), C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
        L.oracle_lattice_coords.argtypes = [C.POINTER(Box), C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_eval_lattice.argtypes = [C.c_void_p, C.POINTER(Box), C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_estimate_bounding_box.argtypes = [C.c_void_p, C.c_float, C.POINTER(Box),
                                                   C.POINTER(C.c_uint64), C.c_int]
        L.oracle_parse_lookup_table.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
        L.oracle_get_surface.restype = C.c_size_t
        L.oracle_get_surface.argtypes = [C.c_void_p, C.POINTER(Box), C.POINTER(ExportConfig), C.c_void_p,
                                         C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(MeshStats), C.c_int]
        L.oracle_gradient_descent.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_retopologize.restype = C.c_size_t
        L.oracle_retopologize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.oracle_render_preview.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_scene_set_shader.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_write_stl.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
        L.oracle_write_ply.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class LookupTable:
    def __init__(self, text):
        self.count = np.zeros(256, dtype=np.int32)
        self.tris = np.zeros((256, 5, 3), dtype=np.int32)
        self.total = lib().oracle_parse_lookup_table(text.encode(), _ptr(self.count), _ptr(self.tris))
        if self.total < 0:
            raise ValueError("malformed lookup table")


class Scene:
    """One compiled scene (scene.txt + buildprocedure.txt [+ scene.cl bank])."""

    def __init__(self, scene_txt, buildprocedure_txt, scene_cl_path=None, arbitrary_data=None):
        L = lib()
        self._h = L.oracle_scene_parse(scene_txt.encode(), buildprocedure_txt.encode())
        self._bank_lib = None
        self._ad = None
        if scene_cl_path is not None:
            so = build_scene_bank(scene_cl_path)
            self._bank_lib = C.CDLL(so)
            fn = C.cast(self._bank_lib.dcsg_cpu_sdf_bank, C.c_void_p)
            L.oracle_scene_set_bank(self._h, fn)
            ad = np.zeros(131072, dtype=np.float32) if arbitrary_data is None else \
                np.ascontiguousarray(arbitrary_data, dtype=np.float32)
            self._ad = ad
            self._bank_lib.dcsg_cpu_set_arbitrary_data.argtypes = [C.c_void_p]
            self._bank_lib.dcsg_cpu_set_arbitrary_data(_ptr(ad))
            L.oracle_scene_set_shader(self._h, C.cast(self._bank_lib.dcsg_cpu_shader_bank, C.c_void_p),
                                      C.cast(self._bank_lib.dcsg_cpu_set_camera, C.c_void_p))

    @classmethod
    def from_dir(cls, d, use_scene_cl=True):
        scl = os.path.join(d, "scene.cl")
        return cls(open(os.path.join(d, "scene.txt")).read(),
                   open(os.path.join(d, "buildprocedure.txt")).read(),
                   scl if (use_scene_cl and os.path.exists(scl)) else None, read_arbitrary_data(d))

    def __del__(self):
        try:
            lib().oracle_scene_free(self._h)
        except Exception:
            pass

    def sdf(self, x, y, z):
        return float(lib().oracle_primary_sdf(self._h, x, y, z))

    def normal(self, x, y, z):
        out = np.zeros(3, dtype=np.float32)
        lib().oracle_get_normal(self._h, x, y, z, _ptr(out))
        return out

    def eval_sdf(self, pts, threads=0):
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
        out = np.empty(len(pts), dtype=np.float32)
        lib().oracle_eval_sdf(self._h, _ptr(pts), len(pts), _ptr(out), threads)
        return out

    def eval_normal(self, pts, threads=0):
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
        out = np.empty((len(pts), 3), dtype=np.float32)
        lib().oracle_eval_normal(self._h, _ptr(pts), len(pts), _ptr(out), threads)
        return out

    def eval_lattice(self, center, diameter, res, z0=0, z1=None, sdf=True, normals=True, threads=0):
        n1 = res + 1
        z1 = n1 if z1 is None else z1
        bb = Box.make(center, diameter)
        s = np.empty((z1 - z0, n1, n1), dtype=np.float32) if sdf else None
        n = np.empty((z1 - z0, n1, n1, 3), dtype=np.float32) if normals else None
        lib().oracle_eval_lattice(self._h, C.byref(bb), res, z0, z1,
                                  _ptr(s) if sdf else None, _ptr(n) if normals else None, threads)
        return s, n

    def render_preview(self, campos, right, up, forward, threads=0):
        """kernel k1 restated: (480, 640, 3) uint8 image."""
        v = [np.ascontiguousarray(a, dtype=np.float32) for a in (campos, right, up, forward)]
        out = np.zeros((480, 640, 3), dtype=np.uint8)
        lib().oracle_render_preview(self._h, _ptr(v[0]), _ptr(v[1]), _ptr(v[2]), _ptr(v[3]), _ptr(out), threads)
        return out

    def estimate_bounding_box(self, bb_half_diameter, threads=0):
        bb = Box()
        cnt = C.c_uint64(0)
        lib().oracle_estimate_bounding_box(self._h, bb_half_diameter, C.byref(bb), C.byref(cnt), threads)
        return list(bb.center), list(bb.diameters), cnt.value

    def get_surface(self, center, diameter, cfg, table, threads=0):
        bb = Box.make(center, diameter)
        out = C.c_void_p()
        st = MeshStats()
        n = lib().oracle_get_surface(self._h, C.byref(bb), C.byref(cfg), _ptr(table.count), _ptr(table.tris),
                                     C.byref(out), C.byref(st), threads)
        arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_float)), shape=(n, 3, 3)).copy() if n else \
            np.zeros((0, 3, 3), dtype=np.float32)
        lib().oracle_free(out)
        return arr, st

    def gradient_descent(self, tris, steps, threads=0):
        tris = np.ascontiguousarray(tris, dtype=np.float32).copy()
        lib().oracle_gradient_descent(self._h, steps, _ptr(tris), tris.shape[0], threads)
        return tris


def retopologize(tris, center, diameter, min_level, grid_level):
    """cms::retopologize as intended (see dcsg_oracle.c: parity unpinned, the reference's is undefined behaviour)."""
    t = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 3, 3)
    c = np.ascontiguousarray(center, dtype=np.float32)
    d = np.ascontiguousarray([diameter] * 3 if np.isscalar(diameter) else diameter, dtype=np.float32)
    hd = (d / np.float32(2.0)).astype(np.float32)   # DesignCSG.cpp:718
    out = C.c_void_p()
    n = lib().oracle_retopologize(_ptr(t), len(t), _ptr(c), _ptr(hd), min_level, grid_level, C.byref(out))
    arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_float)), shape=(n, 3, 3)).copy() if n else \
        np.zeros((0, 3, 3), dtype=np.float32)
    lib().oracle_free(out)
    return arr


def read_arbitrary_data(d):
    """arbitrary_data.hex (131072 LE f32) of a scene directory, or its stored non-zero prefix; None if absent."""
    for name in ("arbitrary_data.hex", "arbitrary_data.prefix.hex"):
        f = os.path.join(d, name)
        if os.path.exists(f):
            a = np.zeros(131072, dtype=np.float32)
            v = np.fromfile(f, dtype="<f4")
            a[:len(v)] = v
            return a
    return None


def parse_export_config(text):
    cfg = ExportConfig()
    if lib().oracle_parse_export_config(text.encode(), C.byref(cfg)) != 0:
        raise ValueError("malformed exportConfig.txt")
    return cfg


def lattice_point(center, diameter, res, ix, iy, iz):
    bb = Box.make(center, diameter)
    out = np.zeros(3, dtype=np.float32)
    lib().oracle_lattice_point(C.byref(bb), res, ix, iy, iz, _ptr(out))
    return out


def lattice_coords(center, diameter, res, p):
    bb = Box.make(center, diameter)
    p = np.ascontiguousarray(p, dtype=np.float32)
    out = np.zeros(3, dtype=np.int64)
    lib().oracle_lattice_coords(C.byref(bb), res, _ptr(p), _ptr(out))
    return out


def write_stl(path, tris):
    tris = np.ascontiguousarray(tris, dtype=np.float32)
    return lib().oracle_write_stl(path.encode(), _ptr(tris), tris.shape[0])


def write_ply(path, tris):
    tris = np.ascontiguousarray(tris, dtype=np.float32)
    return lib().oracle_write_ply(path.encode(), _ptr(tris), tris.shape[0])
