import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
TABLE = os.path.join(ROOT, "designcsg_amd", "data", "lookupTable.txt")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def lookup_table(oracle):
    return oracle.LookupTable(open(TABLE).read())


@pytest.fixture(scope="session")
def built_lib():
    """libdcsg_hip.so, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    from designcsg_amd import build
    build.build(prebuild_scenes=False)
    from designcsg_amd import _lib
    return _lib.lib()


def scene_dir(name):
    return os.path.join(SCENES, name)


def read_scene(name):
    d = scene_dir(name)
    out = {f: open(os.path.join(d, f)).read() for f in ("scene.txt", "buildprocedure.txt", "scene.cl",
                                                         "exportConfig.txt")}
    ad = os.path.join(d, "arbitrary_data.prefix.hex")
    if os.path.exists(ad):
        import numpy as np
        out["arbitrary_data"] = np.fromfile(ad, dtype="<f4")
    return out


def export_config_text(name, levels=None, gd=None):
    """exportConfig.txt of a fixture scene with the octree levels / GD steps overridden."""
    lines = read_scene(name)["exportConfig.txt"].split("\n")
    if levels is not None:
        lines[1], lines[2], lines[3] = (str(v) for v in levels)
    if gd is not None:
        lines[5] = str(gd)
    return "\n".join(lines)
