"""Builds every native artefact of designcsg_amd for gfx950, in-tree under designcsg_amd/_build/.

  dcsg_builtin.hsaco        kernels.hip with the HIP-native built-in brush bank (code object)
  dcsg_kernels_{spec,interp}_{export,rest}.bc          kernels.hip as device bitcode (two kernel sets, each in two
                                                       parts), linked against a scene.cl at load time
  dcsg_kernels_relaxed_{spec,interp}_{export,rest}.bc  the same with the opt-in relaxed arithmetic (dcsg_set_numerics)
  libdcsg_hip.so            the C ABI (include/dcsg.h): host side + comgr JIT
  jitcache/*.hsaco          per-scene code objects compiled ahead of time (optional warm-up)

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container; the outputs travel
to the GPU box with the source tree.
"""
import ctypes
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_build")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
ARCH = "gfx950"
# numerics contract of kernels.hip: no contraction, IEEE divide/sqrt (hipcc default), no fast-math
# -fno-slp-vectorize: packed-f32 VALU ops are slower than the scalar pairs they replace on gfx950
DEVICE_FLAGS = ["--offload-arch=" + ARCH, "--cuda-device-only", "-O3", "-ffp-contract=off", "-fno-slp-vectorize",
                "-std=c++17"]
# A/B builds only (tools/ab_sweep.py on a copy of the tree): extra defines for kernels.hip, e.g. DCSG_BUILD_DEFS="-DDCSG_NO_NT_STORES"
DEVICE_FLAGS += [f for f in os.environ.get("DCSG_BUILD_DEFS", "").split() if f]
LIB = os.path.join(OUT, "libdcsg_hip.so")
RELAXED_SCENES = ("sphere", "design2", "design1")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build step failed: " + cmd[0])


def _embed_cl():
    """scene_prelude.cl / scene_epilogue.cl -> C++ raw string literals for jit.cpp."""
    dst = os.path.join(OUT, "embedded_cl.inc")
    srcs = [os.path.join(CSRC, "scene_prelude.cl"), os.path.join(CSRC, "scene_epilogue.cl")]
    if not _newer(dst, srcs):
        return
    with open(dst, "w") as f:
        for name, path in zip(("kScenePrelude", "kSceneEpilogue"), srcs):
            f.write('static const char %s[] = R"DCSG_CL(%s)DCSG_CL";\n' % (name, open(path).read()))


def build(verbose=False, prebuild_scenes=True):
    os.makedirs(OUT, exist_ok=True)
    kernels = os.path.join(CSRC, "kernels.hip")
    dev_deps = [kernels, os.path.join(CSRC, "dcsg_device_params.h")]

    hsaco = os.path.join(OUT, "dcsg_builtin.hsaco")
    if _newer(hsaco, dev_deps):
        _run([HIPCC] + DEVICE_FLAGS + ["--no-gpu-bundle-output", "-c", kernels, "-o", hsaco])
    # Device bitcode linked against a scene's brush text at load time (csrc/jit.cpp), in two kernel sets so that a cold
    # JIT compiles only what the scene runs: "spec" (per-scene specialised kernels + mesher) and "interp" (the
    # interpreters + mesher + preview); each also with the opt-in relaxed arithmetic (dcsg_set_numerics).
    rebuilt = False
    relaxed_flags = [f for f in DEVICE_FLAGS if f != "-ffp-contract=off"] + ["-ffp-contract=fast"]
    # ... and each set in two parts, so that a cold JIT compiles first what an export launches (csrc/jit.cpp): "export"
    # (stage A, node-wise sweep, mesher, projection, de-duplication) and "rest" (dense sweeps, point evaluation,
    # T-junction repair, test kernels, preview).
    for set_name, set_def in (("spec", "-DDCSG_SET_SPEC"), ("interp", "-DDCSG_SET_INTERP")):
        for part, part_def in (("export", "-DDCSG_PART_EXPORT"), ("rest", "-DDCSG_PART_REST")):
            for prefix, flags, extra in (("dcsg_kernels_", DEVICE_FLAGS, []), ("dcsg_kernels_relaxed_", relaxed_flags, ["-DDCSG_RELAXED"])):
                bc = os.path.join(OUT, prefix + set_name + "_" + part + ".bc")
                if _newer(bc, dev_deps):
                    _run([HIPCC] + flags + ["-fgpu-rdc", "-emit-llvm", "-DDCSG_EXTERN_BANK", set_def, part_def] + extra + ["-c", kernels, "-o", bc])
                    rebuilt = True
    if rebuilt:
        # every cached scene module was linked against the old kernel bitcode (its hash is part of the key):
        # drop them so that stale code objects do not pile up (the fixture scenes are rebuilt below)
        for f in glob.glob(os.path.join(OUT, "jitcache", "*.hsaco")):
            os.remove(f)
    for stale in ("dcsg_kernels_extern.bc", "dcsg_kernels_relaxed.bc", "dcsg_kernels_spec.bc", "dcsg_kernels_interp.bc",
                  "dcsg_kernels_relaxed_spec.bc", "dcsg_kernels_relaxed_interp.bc"):
        if os.path.exists(os.path.join(OUT, stale)):
            os.remove(os.path.join(OUT, stale))

    _embed_cl()
    host_src = [os.path.join(CSRC, f) for f in ("dcsg_api.cpp", "scene.cpp", "jit.cpp", "specialise.cpp")]
    host_deps = host_src + glob.glob(os.path.join(CSRC, "*.h*")) + \
        [os.path.join(OUT, "embedded_cl.inc"), os.path.join(HERE, "..", "include", "dcsg.h")]
    if _newer(LIB, host_deps):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
              "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include"), "-I" + OUT, "-I" + CSRC]
             + host_src + ["-o", LIB, "-L" + os.path.join(ROCM, "lib"), "-lamdhip64", "-ldl", "-lpthread",
                           "-Wl,-rpath," + os.path.join(ROCM, "lib")])
    if prebuild_scenes:
        prebuild_jit_cache(verbose)
    return LIB


def prebuild_jit_cache(verbose=False):
    """Compile the code objects the fixture scenes need (brush text + specialised build procedure, and the
    un-specialised variant the interpreter tests use) into the JIT cache.  No GPU needed."""
    lib = ctypes.CDLL(LIB)
    lib.dcsg_jit_prebuild_scene.argtypes = [ctypes.c_char_p] * 4 + [ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
    scenes = sorted(glob.glob(os.path.join(HERE, "..", "tests", "golden", "scenes", "*", "scene.cl")))
    for path in scenes:
        d = os.path.dirname(path)
        rd = lambda n: open(os.path.join(d, n), "rb").read()
        for spec in ("48", "0"):
            os.environ["DCSG_SPECIALISE"] = spec
            log = ctypes.create_string_buffer(1 << 16)
            cached = ctypes.c_int(0)
            rc = lib.dcsg_jit_prebuild_scene(rd("scene.txt"), rd("buildprocedure.txt"), rd("scene.cl"), log, len(log),
                                             ctypes.byref(cached))
            if rc != 0:
                raise RuntimeError("JIT of %s failed:\n%s" % (path, log.value.decode(errors="replace")))
            if verbose:
                print("jit", os.path.relpath(d), "specialise<=" + spec, "cached" if cached.value else "compiled")
    os.environ.pop("DCSG_SPECIALISE", None)
    # the relaxed-numerics modules of the scenes tests/test_gpu_relaxed.py and bench.py load
    lib.dcsg_jit_prebuild_scene_numerics.argtypes = [ctypes.c_char_p] * 3 + [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t,
                                                                             ctypes.POINTER(ctypes.c_int)]
    for name in RELAXED_SCENES:
        d = os.path.join(HERE, "..", "tests", "golden", "scenes", name)
        rd = lambda n: open(os.path.join(d, n), "rb").read()
        log = ctypes.create_string_buffer(1 << 16)
        rc = lib.dcsg_jit_prebuild_scene_numerics(rd("scene.txt"), rd("buildprocedure.txt"), rd("scene.cl"), 1, log, len(log), None)
        if rc != 0:
            raise RuntimeError("relaxed JIT of %s failed:\n%s" % (name, log.value.decode(errors="replace")))


if __name__ == "__main__":
    build(verbose=True)
    print(LIB)
