"""Runs the REFERENCE's own kernel `k2` (k2.cl:234-280) on the MI355X.

TEST INFRASTRUCTURE ONLY (the checker): used by tests/, tests/golden/make_k2_vectors.py and the
`reference_kernel_same_box` leg of bench.py's cpu_baseline -- never by designcsg_amd.

`make -C oracle ref-k2` compiles the reference's kernel text (k2.cl + the scene's scene.cl = k2build.cl,
DesignCSG.cpp:546) from where it lies under /root/reference into oracle/_ref/k2_<scene>_<flavour>.hsaco with the
image's OpenCL front-end and ROCm's OpenCL device library -- what clBuildProgram (Evaluator.cpp:45-112,
Utils.cpp:78-81) produces on this hardware.  This module is the reference's host side around it, restated over
the HIP module API (the box's OpenCL ICD is not needed):

  * loadScene's banks (DrawPane.cpp:267-371): shape ids (uchar), position / right / up / forward (3 floats per
    object), 4 ints per build step, the two one-element count buffers;
  * Evaluator::build's 12 kernel arguments in their order (Evaluator.cpp:90-105);
  * eval_sdf_at_points / eval_normal_at_points (Evaluator.cpp:117-211): eval_types[0] = 0 / 1, batches of at most
    MAX_EVAL_POINTS = 16 777 216 (Evaluator.h:16), global size = points, LOCAL SIZE 1 (Evaluator.cpp:149-152,195-198);
  * setArbitraryData's 131 072-float buffer (Evaluator.cpp:213-225).

The reference allocates its banks for 512 objects / 256 steps (DrawPane.h:14-15); the kernel itself has no such
limit, so the buffers here are sized by the scene and flat256 / grouped256 (257 objects, 514+ steps) run through the
unchanged kernel text.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
REFERENCE = "/root/reference/master"
MAX_EVAL_POINTS = 16777216          # Evaluator.h:16
ARBITRARY_DATA_POINTS = 131072      # Evaluator.h:17
FLAVOURS = ("default", "strict")
# Scenes whose files the reference's own kernel can run: everything captured from the reference front-end except
# global_state (its brushes write program-scope __global variables that a later call reads back: a data race as soon
# as two work-items run at once, which is what a GPU does) and the smooth-join extension scenes (opcodes k2.cl's
# switch ignores).
SCENES = ("sphere", "design1", "design2", "mixed_groups", "flat256", "grouped256", "prefabs", "const_table", "ad_table", "userbrush128")


def hsaco_path(scene, flavour="strict"):
    return os.path.join(REF_DIR, "k2_%s_%s.hsaco" % (scene, flavour))


def build_ref_k2(scenes=SCENES, scenes_root=None, force=False):
    """Compile the code objects where the reference tree is present (build container); elsewhere report what
    is already there.  Returns the list of scenes whose code objects exist."""
    scenes_root = scenes_root or os.path.join(HERE, "..", "tests", "golden", "scenes")
    have = []
    for name in scenes:
        scl = os.path.abspath(os.path.join(scenes_root, name, "scene.cl"))
        outs = [hsaco_path(name, f) for f in FLAVOURS]
        if os.path.isdir(REFERENCE) and os.path.exists(scl):
            deps = [scl, os.path.join(REFERENCE, "k2.cl"), os.path.join(HERE, "Makefile")]
            stale = force or any(not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(d) for d in deps)
                                 for o in outs)
            if stale:
                subprocess.run(["make", "-C", HERE, "-s", "ref-k2", "NAME=" + name, "SCENE_CL=" + scl], check=True)
        if all(os.path.exists(o) for o in outs):
            have.append(name)
    return have


_hip = None


def hip():
    global _hip
    if _hip is None:
        try:
            import torch  # noqa: F401  (one HIP runtime per process: bind to the copy torch loaded, as designcsg_amd/_lib.py does)
        except ImportError:
            pass
        L = C.CDLL("libamdhip64.so.7")
        vp, sz = C.c_void_p, C.c_size_t
        L.hipGetErrorString.restype = C.c_char_p
        L.hipGetErrorString.argtypes = [C.c_int]
        L.hipSetDevice.argtypes = [C.c_int]
        L.hipModuleLoadData.argtypes = [C.POINTER(vp), vp]
        L.hipModuleUnload.argtypes = [vp]
        L.hipModuleGetFunction.argtypes = [C.POINTER(vp), vp, C.c_char_p]
        L.hipMalloc.argtypes = [C.POINTER(vp), sz]
        L.hipFree.argtypes = [vp]
        L.hipMemcpy.argtypes = [vp, vp, sz, C.c_int]
        L.hipMemset.argtypes = [vp, C.c_int, sz]
        L.hipModuleLaunchKernel.argtypes = [vp] + [C.c_uint] * 6 + [C.c_uint, vp, C.POINTER(vp), C.POINTER(vp)]
        L.hipDeviceSynchronize.argtypes = []
        L.hipEventCreate.argtypes = [C.POINTER(vp)]
        L.hipEventRecord.argtypes = [vp, vp]
        L.hipEventSynchronize.argtypes = [vp]
        L.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), vp, vp]
        L.hipEventDestroy.argtypes = [vp]
        _hip = L
    return _hip


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError("%s: HIP error %d (%s)" % (what, rc, hip().hipGetErrorString(rc).decode()))


H2D, D2H = 1, 2


class _Buf:
    def __init__(self, nbytes):
        self.p = C.c_void_p()
        self.nbytes = max(int(nbytes), 4)
        _ck(hip().hipMalloc(C.byref(self.p), self.nbytes), "hipMalloc")

    def upload(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        if a.nbytes:
            _ck(hip().hipMemcpy(self.p, a.ctypes.data_as(C.c_void_p), a.nbytes, H2D), "hipMemcpy h2d")

    def download(self, a):
        assert a.flags.c_contiguous and a.nbytes <= self.nbytes
        if a.nbytes:
            _ck(hip().hipMemcpy(a.ctypes.data_as(C.c_void_p), self.p, a.nbytes, D2H), "hipMemcpy d2h")

    def free(self):
        if self.p:
            hip().hipFree(self.p)
            self.p = C.c_void_p()


def parse_scene_txt(text):
    """loadScene's parse of scene.txt (DrawPane.cpp:279-345): %d %d then 12 x %f per line."""
    rows = [ln.split() for ln in text.split("\n") if ln.strip()]
    shape = np.array([int(r[0]) for r in rows], dtype=np.uint8)
    f = np.array([[np.float32(x) for x in r[2:14]] for r in rows], dtype=np.float32).reshape(len(rows), 12)
    return shape, f[:, 0:3].copy(), f[:, 3:6].copy(), f[:, 6:9].copy(), f[:, 9:12].copy()


def parse_build_procedure(text):
    """DrawPane.cpp:355-371: `%d %d %d %d` per line."""
    rows = [[int(v) for v in ln.split()[:4]] for ln in text.split("\n") if ln.strip()]
    return np.array(rows, dtype=np.int32).reshape(len(rows), 4)


class RefK2:
    """One scene through the reference's kernel.  `local_size` defaults to the reference's 1."""

    def __init__(self, scene_dir, flavour="strict", device=0, scene_name=None, max_points=1 << 20):
        name = scene_name or os.path.basename(os.path.normpath(scene_dir))
        path = hsaco_path(name, flavour)
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (make -C oracle ref-k2 in the build container)")
        L = hip()
        _ck(L.hipSetDevice(device), "hipSetDevice")
        self._image = open(path, "rb").read()
        self.module = C.c_void_p()
        _ck(L.hipModuleLoadData(C.byref(self.module), self._image), "hipModuleLoadData " + path)
        self.fn = C.c_void_p()
        _ck(L.hipModuleGetFunction(C.byref(self.fn), self.module, b"k2"), "hipModuleGetFunction k2")
        rd = lambda n: open(os.path.join(scene_dir, n)).read()
        shape, pos, right, up, fwd = parse_scene_txt(rd("scene.txt"))
        prog = parse_build_procedure(rd("buildprocedure.txt"))
        self.num_objects, self.num_steps = len(shape), len(prog)
        ad = np.zeros(ARBITRARY_DATA_POINTS, dtype=np.float32)
        for f in ("arbitrary_data.hex", "arbitrary_data.prefix.hex"):
            if os.path.exists(os.path.join(scene_dir, f)):
                v = np.fromfile(os.path.join(scene_dir, f), dtype="<f4")
                ad[:len(v)] = v
                break
        self.max_points = min(int(max_points), MAX_EVAL_POINTS)
        mk = lambda a: (_Buf(a.nbytes), a)
        host = [shape, pos, right, up, fwd, np.array([self.num_objects], np.int32), prog,
                np.array([self.num_steps], np.int32), ad]
        self._scene = []
        for a in host:
            b = _Buf(a.nbytes)
            b.upload(a)
            self._scene.append(b)
        self.eval_points = _Buf(12 * self.max_points)
        self.eval_out = _Buf(12 * self.max_points)
        self.eval_types = _Buf(4)
        self.last_kernel_ms = 0.0
        self._ev = [C.c_void_p(), C.c_void_p()]
        for e in self._ev:
            _ck(L.hipEventCreate(C.byref(e)), "hipEventCreate")

    def close(self):
        if getattr(self, "module", None):
            for b in self._scene + [self.eval_points, self.eval_out, self.eval_types]:
                b.free()
            for e in self._ev:
                hip().hipEventDestroy(e)
            hip().hipModuleUnload(self.module)
            self.module = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _launch(self, n, local_size):
        L = hip()
        s = self._scene
        order = [self.eval_points, self.eval_out, self.eval_types, s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8]]
        vals = [C.c_void_p(b.p.value) for b in order]
        params = (C.c_void_p * 12)(*[C.cast(C.pointer(v), C.c_void_p) for v in vals])
        assert n % local_size == 0
        _ck(L.hipEventRecord(self._ev[0], None), "hipEventRecord")
        _ck(L.hipModuleLaunchKernel(self.fn, n // local_size, 1, 1, local_size, 1, 1, 0, None, params, None),
            "hipModuleLaunchKernel k2")
        _ck(L.hipEventRecord(self._ev[1], None), "hipEventRecord")
        _ck(L.hipDeviceSynchronize(), "k2")
        ms = C.c_float()
        _ck(L.hipEventElapsedTime(C.byref(ms), self._ev[0], self._ev[1]), "hipEventElapsedTime")
        self.last_kernel_ms += ms.value

    def _eval(self, points, eval_type, width, local_size):
        p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        out = np.empty((len(p), width), dtype=np.float32)
        self.eval_types.upload(np.array([eval_type], dtype=np.int32))
        self.last_kernel_ms = 0.0
        cursor = 0
        while cursor < len(p):
            amt = min(self.max_points, len(p) - cursor)
            pad = (-amt) % local_size
            chunk = p[cursor:cursor + amt]
            if pad:
                chunk = np.concatenate([chunk, np.zeros((pad, 3), np.float32)])
            self.eval_points.upload(chunk)
            self._launch(amt + pad, local_size)
            self.eval_out.download(out[cursor:cursor + amt])
            cursor += amt
        return out

    def eval_sdf_at_points(self, points, local_size=1):
        return self._eval(points, 0, 1, local_size).reshape(-1)

    def eval_normal_at_points(self, points, local_size=1):
        return self._eval(points, 1, 3, local_size)

    # the reference's host loops around the evaluator, restated for the pins of stage A and the projection
    def estimate_bounding_box(self, bb_half_diameter, local_size=1):
        """Export stage A (DesignCSG.cpp:668-712) over this evaluator: returns (center[3], diameter, interior)."""
        cd = np.float32(bb_half_diameter) / np.float32(256.0)
        i = np.arange(-128, 128, dtype=np.int32)
        c = (-cd / np.float32(2.0)) + i.astype(np.float32) * cd           # (-cd/2) + i*cd in float
        pts = np.stack(np.meshgrid(c, c, c, indexing="ij"), -1).reshape(-1, 3)   # x outer, z inner
        s = self.eval_sdf_at_points(pts, local_size)
        inside = s < cd
        lo = np.zeros(3, np.float32)
        hi = np.zeros(3, np.float32)
        if inside.any():
            q = pts[inside]
            lo = np.minimum(lo, q.min(0))
            hi = np.maximum(hi, q.max(0))
        center = ((lo + hi) / np.float32(2.0)).astype(np.float32)
        diameter = np.float32((hi - lo).max())
        return [float(v) for v in center], float(diameter), int(inside.sum())

    def gradient_descent(self, vertices, steps, local_size=1):
        """cms::performGradientDescent (mesh.hpp:531-593): steps x { s = sdf(p); n = normal(p); p += n * (-s) }."""
        p = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3).copy()
        for _ in range(steps):
            s = self.eval_sdf_at_points(p, local_size)
            n = self.eval_normal_at_points(p, local_size)
            p = (p + n * (-s)[:, None]).astype(np.float32)
        return p
