"""Host threads without a GPU: the background build of dcsg_load_scene (csrc/dcsg_api.cpp: RestJob) through the
device-free hook dcsg_debug_background_build -- two scenes loaded back to back with a COLD cache, the first load returning
while its rest part is still being compiled (what an early error return of dcsg_load_scene leaves behind), the second
one having to join it; then a context destroyed with a build in flight.  The same sequences run under ThreadSanitizer in
tools/sanitize_host.sh --thread (tools/tsan_host.cpp; report: profiles/r04_tsan_report.txt)."""
import ctypes as C
import glob
import os

from conftest import read_scene


def test_two_cold_loads_back_to_back(built_lib, tmp_path, monkeypatch):
    L = built_lib
    cache = tmp_path / "cache"
    monkeypatch.setenv("DCSG_JIT_CACHE", str(cache))
    slot = C.c_void_p()
    log = C.create_string_buffer(1 << 16)

    def load(name, mode):
        s = read_scene(name)
        rc = L.dcsg_debug_background_build(C.byref(slot), s["scene.txt"].encode(), s["buildprocedure.txt"].encode(),
                                           s["scene.cl"].encode(), mode, log, len(log))
        assert rc == 0, log.value.decode(errors="replace")

    load("const_table", 1)              # returns with the rest part still compiling
    # the caller changes the JIT's switches before its next call: the running build took its copy when it started
    monkeypatch.setenv("DCSG_JIT_EXTRA_OPTS", "-DCHURN=1")
    monkeypatch.delenv("DCSG_JIT_EXTRA_OPTS")
    load("sphere", 0)                   # joins the first build, compiles both parts of the second scene
    objs = sorted(glob.glob(str(cache / "*.hsaco")))
    assert len(objs) == 4, objs         # export + rest part of both scenes, each written whole (atomic rename)
    assert all(os.path.getsize(f) > 10000 and open(f, "rb").read(4) == b"\x7fELF" for f in objs)
    load("const_table", 1)              # from the cache now: no background build at all
    load("prefabs", 1)                  # cold again, abandoned ...
    assert L.dcsg_debug_background_build(C.byref(slot), None, None, None, 2, log, len(log)) == 0   # ... and destroyed: joins
    assert slot.value is None
    assert len(glob.glob(str(cache / "*.hsaco"))) == 6
    # malformed input is reported, not started
    slot2 = C.c_void_p()
    assert L.dcsg_debug_background_build(C.byref(slot2), b"garbage", b"0 0 0 0", b"", 0, log, len(log)) != 0
    assert L.dcsg_debug_background_build(C.byref(slot2), None, None, None, 2, log, len(log)) == 0
