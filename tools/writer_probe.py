#!/usr/bin/env python3
"""Where do the file writers' milliseconds go?  Writes N random triangles as .stl and .ply (whole-file writers of the C ABI,
no GPU) into each directory given, six times each, and prints the times: the first call creates the file, the others
overwrite it in place.  A tmpfs directory (/dev/shm) shows the cost without a disk file system's page-fault path."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from designcsg_amd import _lib

def main():
    n = int(os.environ.get("N_TRIS", 1600000))
    dirs = sys.argv[1:] or ["/tmp", "/dev/shm"]
    lib = _lib.lib()
    tris = np.random.default_rng(1).uniform(-5, 5, (n, 9)).astype(np.float32)
    p = tris.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    for d in dirs:
        for name, fn in (("stl", lib.dcsg_write_stl), ("ply", lib.dcsg_write_ply)):
            path = os.path.join(d, "probe_%d.%s" % (os.getpid(), name))
            ts = []
            for _ in range(6):
                t0 = time.perf_counter()
                rc = fn(path.encode(), p, n)
                ts.append((time.perf_counter() - t0) * 1e3)
                assert rc == 0
            size = os.path.getsize(path)
            os.unlink(path)
            print("%-28s %s %6.1f MB  first %.1f ms, then %s ms  (%.1f GB/s)" % (d, name, size / 1e6, ts[0], " ".join("%.1f" % t for t in ts[1:]),
                                                                            size / 1e6 / min(ts[1:])))

if __name__ == "__main__":
    main()
