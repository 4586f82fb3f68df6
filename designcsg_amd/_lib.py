"""ctypes binding of libdcsg_hip.so (include/dcsg.h).

The library is the product: if it is missing this module raises, it never falls back to a
CPU implementation.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libdcsg_hip.so")

MAX_LEVELS = 16
ARBITRARY_DATA_POINTS = 131072

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_BUILD, ERR_NO_SCENE, ERR_IO = 0, -1, -2, -3, -4, -5, -6
EXPORT_DENSE_NORMALS = 1
EXPORT_SPARSE_NORMALS = 2
EXPORT_INDEXED_PLY = 4     # opt-in writers beyond the reference's, include/dcsg.h
EXPORT_STL_NORMALS = 8
EXPORT_DENSE_SWEEP = 16
EXPORT_EQUAL_SLABS = 64   # equal Z slabs instead of the work-balanced cuts (dcsg_export_slab / _multi)
EXPORT_RETOPO = 32        # opt-in intended-semantics retopologize between mesher and projection


class ExportConfig(C.Structure):
    _fields_ = [("bounding_box_half_diameter", C.c_float),
                ("min_level", C.c_int32), ("max_level", C.c_int32), ("grid_level", C.c_int32),
                ("complex_surface_threshold", C.c_float),
                ("gradient_descent_steps", C.c_int32),
                ("cache_subdivision", C.c_int32), ("queries_before_gc", C.c_int32),
                ("queries_before_free", C.c_int32)]


class MeshStats(C.Structure):
    _fields_ = [(n, C.c_uint64 * MAX_LEVELS) for n in ("nodes", "culled", "subdivided", "leaves", "triangles")]

    def as_dict(self):
        return {n: [int(v) for v in getattr(self, n)] for n, _ in self._fields_}


class ExportStats(C.Structure):
    _fields_ = [("box_center", C.c_float * 3), ("box_diameter", C.c_float),
                ("interior_points", C.c_uint64), ("lattice_points", C.c_uint64), ("triangles", C.c_uint64),
                ("mesh", MeshStats),
                ("t_bbox", C.c_double), ("t_sweep", C.c_double), ("t_mesh", C.c_double),
                ("t_project", C.c_double), ("t_download", C.c_double), ("t_write_stl", C.c_double),
                ("t_write_ply", C.c_double), ("t_total", C.c_double), ("sweep_kernel_ms", C.c_double),
                ("swept_points", C.c_uint64), ("swept_nodes", C.c_uint64), ("slab_z_lo", C.c_int32), ("slab_z_hi", C.c_int32)]


# name -> (restype, argtypes); every prototype of include/dcsg.h
PROTOTYPES = {
    "dcsg_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "dcsg_destroy": (None, [C.c_void_p]),
    "dcsg_last_error": (C.c_char_p, [C.c_void_p]),
    "dcsg_load_scene": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "dcsg_set_arbitrary_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "dcsg_set_numerics": (C.c_int, [C.c_void_p, C.c_int]),
    "dcsg_scene_info": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int)] * 6),
    "dcsg_debug_scene_flags": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "dcsg_eval_sdf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dcsg_eval_normal": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dcsg_eval_sdf_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "dcsg_eval_normal_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "dcsg_sweep_lattice_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                            C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dcsg_eval_lattice": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int,
                                    C.c_int, C.c_void_p, C.c_void_p]),
    "dcsg_estimate_bounding_box": (C.c_int, [C.c_void_p, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                             C.POINTER(C.c_uint64)]),
    "dcsg_set_lookup_table": (C.c_int, [C.c_void_p, C.c_char_p]),
    "dcsg_mesh_lattice_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_float, C.POINTER(ExportConfig),
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(C.c_uint64), C.POINTER(MeshStats), C.c_void_p]),
    "dcsg_retopologize": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "dcsg_upload_triangles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "dcsg_project_triangles": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "dcsg_project_vertices_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "dcsg_triangles_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "dcsg_download_triangles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "dcsg_copy_triangles_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "dcsg_render_preview": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dcsg_write_stl": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint64]),
    "dcsg_write_ply": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint64]),
    "dcsg_write_stl_range": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint]),
    "dcsg_write_ply_range": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint]),
    "dcsg_write_stl_normals": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint64]),
    "dcsg_write_ply_indexed": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "dcsg_parse_export_config": (C.c_int, [C.c_char_p, C.POINTER(ExportConfig)]),
    "dcsg_export": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint, C.POINTER(ExportStats)]),
    "dcsg_export_slab": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_uint, C.POINTER(ExportStats)]),
    "dcsg_export_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint, C.POINTER(ExportStats)]),
    "dcsg_write_slab": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint]),
    "dcsg_load_profile": (C.c_char_p, [C.c_void_p, C.c_int]),
    "dcsg_last_sweep_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "dcsg_set_sweep_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "dcsg_compile_program": (C.c_int, [C.c_char_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    "dcsg_expand_lookup_table": (C.c_int, [C.c_char_p, C.c_void_p, C.POINTER(C.c_int)]),
    "dcsg_angle_q_star": (C.c_float, [C.c_float]),
    "dcsg_debug_slab_cuts": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dcsg_debug_cells": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "dcsg_debug_sqrt_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "dcsg_debug_normal_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "dcsg_debug_background_build": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]),
    "dcsg_jit_prebuild_scene": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t,
                                          C.POINTER(C.c_int)]),
    "dcsg_jit_prebuild": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "dcsg_jit_prebuild_scene_numerics": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t,
                                                   C.POINTER(C.c_int)]),
}

_lib = None


class DcsgError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libdcsg_hip error %d: %s" % (code, message))
        self.code = code
        self.message = message


def lib():
    """Load libdcsg_hip.so.  Raises if it has not been built: there is no other backend."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(designcsg_amd has no CPU fallback)" % LIB_PATH)
        # One HIP runtime per process: PyTorch bundles its own libamdhip64 / libhsa-runtime64 under the
        # same SONAMEs as /opt/rocm.  Importing torch first makes libdcsg_hip bind to that copy, so that
        # torch tensors, streams and RCCL share one runtime with our kernels.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(ctx, rc):
    if rc != 0:
        msg = lib().dcsg_last_error(ctx)
        raise DcsgError(rc, msg.decode(errors="replace") if msg else "")
    return rc
