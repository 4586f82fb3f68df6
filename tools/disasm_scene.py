#!/usr/bin/env python3
"""Disassembles one kernel of a fixture scene's code object (no GPU needed):
    python tools/disasm_scene.py sphere dcsg_sweep_sp [--spec 48] > /tmp/k.s
Compiles the scene into a scratch JIT cache, runs llvm-objdump on the result and prints the kernel's ISA with a
per-class instruction count (static counts; loops are not weighted)."""
import collections
import ctypes
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.environ.get("DCSG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def main():
    scene, kernel = sys.argv[1], sys.argv[2]
    spec = sys.argv[sys.argv.index("--spec") + 1] if "--spec" in sys.argv else "48"
    d = os.path.join(ROOT, "tests", "golden", "scenes", scene)
    rd = lambda n: open(os.path.join(d, n), "rb").read()
    with tempfile.TemporaryDirectory() as tmp:
        os.environ["DCSG_JIT_CACHE"] = tmp
        os.environ["DCSG_SPECIALISE"] = spec
        lib = ctypes.CDLL(os.path.join(ROOT, "designcsg_amd", "_build", "libdcsg_hip.so"))
        lib.dcsg_jit_prebuild_scene.argtypes = [ctypes.c_char_p] * 4 + [ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
        log = ctypes.create_string_buffer(1 << 16)
        rc = lib.dcsg_jit_prebuild_scene(rd("scene.txt"), rd("buildprocedure.txt"), rd("scene.cl"), log, len(log), None)
        if rc:
            sys.exit(log.value.decode())
        objs = glob.glob(os.path.join(tmp, "*.hsaco"))
        if not objs:
            objs = [os.path.join(ROOT, "designcsg_amd", "_build", "dcsg_builtin.hsaco")]  # scene needs no JIT
        # a scene module is two code objects (export part, rest part): look for the kernel in each
        txt = "".join(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", o], capture_output=True, text=True).stdout for o in objs)
    m = re.search(r"^[0-9a-f]+ <%s>:\n(.*?)(?=^[0-9a-f]+ <|\Z)" % re.escape(kernel), txt, re.S | re.M)
    if not m:
        sys.exit("kernel %s not found" % kernel)
    body = m.group(1)
    cls = collections.Counter()
    for line in body.splitlines():
        t = line.split()
        if not t or t[0].endswith(":") or t[0].startswith("//"):
            continue
        op = t[0]
        k = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
             else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        cls[k] += 1
    print(body)
    print("// static instruction counts:", dict(cls), file=sys.stderr)


if __name__ == "__main__":
    main()
