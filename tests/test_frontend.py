"""Own design API (designcsg_amd/frontend) against fixtures captured from the reference Python
(tests/golden/make_scene_fixtures.py): the compiled scene must come out byte for byte."""
import hashlib
import json
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT, SCENES

DESIGNS = os.path.join(ROOT, "tests", "designs")
MANIFEST = json.load(open(os.path.join(SCENES, "MANIFEST.json")))
FILES = ["scene.txt", "buildprocedure.txt", "scene.cl", "exportConfig.txt"]


def run(script, workdir):
    subprocess.run([sys.executable, "-m", "designcsg_amd.run", script, "--workdir", str(workdir)], cwd=ROOT,
                   check=True, stdout=subprocess.DEVNULL)


@pytest.mark.parametrize("name", ["sphere", "mixed_groups", "flat256", "grouped256", "ad_table", "const_table", "global_state", "userbrush128"])
def test_own_design_scripts_reproduce_reference_output(name, tmp_path):
    run(os.path.join(DESIGNS, name + ".py"), tmp_path)
    for f in FILES:
        want = open(os.path.join(SCENES, name, f), "rb").read()
        got = (tmp_path / f).read_bytes()
        assert got == want, "%s/%s differs from the reference capture" % (name, f)
    ad = (tmp_path / "arbitrary_data.hex").read_bytes()
    assert len(ad) == MANIFEST[name]["arbitrary_data_bytes"] == 524288
    assert hashlib.sha256(ad).hexdigest() == MANIFEST[name]["arbitrary_data_sha256"]


@pytest.mark.parametrize("name", ["smooth_blobs", "smooth256"])
def test_extension_fixtures_are_what_the_front_end_writes(name, tmp_path):
    """Smooth-join scenes (extension, tests/golden/make_extension_fixtures.py) are compiled by this front-end;
    the committed fixtures must be its current output."""
    run(os.path.join(DESIGNS, name + ".py"), tmp_path)
    for f in FILES:
        assert (tmp_path / f).read_bytes() == open(os.path.join(SCENES, name, f), "rb").read(), f


def _bodies(scene_cl, kind):
    """Brush / material bodies of a generated scene.cl, in bank order."""
    pat = r"float sd(\d+)\( float3 v\)\{\n\n            (.*?)\n\n        \}" if kind == "sd" else \
        r"float3 shader(\d+) \(float3 gv, float3 lv, float3 n\)\{\n\n            (.*?)\n\n        \}"
    return [m.group(2) for m in re.finditer(pat, scene_cl, re.S)]


def test_design1_equivalent_reproduces_capture(tmp_path):
    """Design1 (reference Designs/Design1.py, not copied): a script with the same calls, written here,
    must compile to the captured files."""
    script = tmp_path / "d1.py"
    script.write_text(
        "from DesignCSG import *\nfrom designlibrary import *\nimport numpy as np\n"
        "s = define_brush(body=''' \\n\\treturn length(v)-0.5;\\n''')\n"
        "b = define_brush(body=''' \\n\\tv= fabs(v);\\n\\treturn T_max(T_max(v.x-0.5,v.y-0.5),v.z-0.5);\\n''')\n"
        "tf = lambda p, sc: Transform.initial(position=np.array(p), yaw=-PI/2, pitch=0, roll=0, scale=np.array(sc))\n"
        "draw(s, tf([0.0,-0.0,0.0], [1.25]*3))\n"
        "draw(b, tf([0.0,-0.0,0.0], [0.95]*3))\n"
        "for _x,_y,_z in np.ndindex((3,3,3)):\n"
        "    x,y,z = _x-1,_y-1,_z-1\n"
        "    if abs(x)+abs(y)+abs(z)==3:\n"
        "        erase(s, tf([x,y,z], 2.15*np.array([1.0,1.0,1.0])))\n"
        "setExportConfig(boundingBoxHalfDiameter=2.0, minimumOctreeLevel=5, maximumOctreeLevel=7, gridLevel=8,\n"
        "    complexSurfaceThreshold=np.pi/2.0*0.5, gradientDescentSteps=50, cacheSubdivision=16,\n"
        "    queriesBeforeGC=512, queriesBeforeFree=4096)\n"
        "commit()\n")
    wd = tmp_path / "wd"
    run(str(script), wd)
    for f in FILES:
        assert (wd / f).read_bytes() == open(os.path.join(SCENES, "design1", f), "rb").read(), f


def test_design2_scene_cl_reassembles_from_its_parts(tmp_path):
    """Design2's brush text lives only in the captured scene.cl (data).  Feeding its parts (defines, helper
    block, brush bodies) back through the own compiler must reproduce every captured file byte for byte."""
    cap = open(os.path.join(SCENES, "design2", "scene.cl")).read()
    sd = _bodies(cap, "sd")
    assert len(sd) == 7
    # text between the AD block and the first brush = defines + "\n\n        " + helpers
    head = cap[:cap.index("\n        float sd0( float3 v){")]
    m = re.match(r"\n        \n\n\n        (.*)\n\n        (.*)\n\n        $", head, re.S)
    assert m, "unexpected scene.cl layout"
    defines_joined, aux_joined = m.group(1), m.group(2)
    assert aux_joined.startswith(" \n")
    script = tmp_path / "d2.py"
    script.write_text(
        "from DesignCSG import *\nimport numpy as np\nimport json\n"
        "parts = json.load(open(%r))\n"
        "compiler.preprocessor_defines.append(parts['defines'])\n"
        "define_auxillary_function(function=parts['aux'])\n"
        "hil = define_brush(body=parts['sd'][5])\nbase = define_brush(body=parts['sd'][6])\n"
        "draw(hil, Transform.initial(position=np.array([0.0,0.0,0.0]), yaw=np.pi/4, pitch=np.pi/4, roll=np.pi/4, scale=np.array([1.0,1.0,1.0])))\n"
        "draw(base, Transform.initial(position=np.array([0.0,-np.sqrt(3*0.25)+0.0075,0.0]), yaw=0.0, pitch=0.0, roll=0.0, scale=np.array([1.0,1.0,1.0])))\n"
        "setExportConfig(boundingBoxHalfDiameter=2.0, minimumOctreeLevel=6, maximumOctreeLevel=8, gridLevel=9,\n"
        "    complexSurfaceThreshold=np.pi/2.0*0.5, gradientDescentSteps=50, cacheSubdivision=16,\n"
        "    queriesBeforeGC=512, queriesBeforeFree=4096)\n"
        "commit()\n" % str(tmp_path / "parts.json"))
    json.dump({"defines": defines_joined, "aux": aux_joined[2:], "sd": sd}, open(tmp_path / "parts.json", "w"))
    wd = tmp_path / "wd"
    run(str(script), wd)
    for f in FILES:
        assert (wd / f).read_bytes() == open(os.path.join(SCENES, "design2", f), "rb").read(), f


def test_transform_conventions():
    sys.path.insert(0, os.path.join(ROOT, "designcsg_amd", "frontend"))
    try:
        import importlib
        sc = importlib.import_module("scenecompiler")
        import numpy as np
        T = sc.Transform
        m = T.initial(np.array([1.0, 2.0, 3.0]), 0.3, -0.2, 0.1, np.array([2.0, 3.0, 4.0]))
        assert np.allclose(m[:3, 3], [1, 2, 3])                       # column-vector convention
        assert np.allclose(np.linalg.norm(m[:3, :3], axis=0), [2, 3, 4])  # axes are columns
        assert np.allclose(T.reciprocal_vector(np.array([0.0, 2.0, 0.0])), [0, 0.5, 0])
        assert np.allclose(sc.matmul(T.eulerY(0.3), T.eulerX(0.2), T.eulerZ(0.1)), T.rotation(0.3, 0.2, 0.1))
        with pytest.raises(TypeError):
            sc.matmul(m)
    finally:
        sys.path.pop(0)
        for mod in ("scenecompiler", "DesignCSG", "designlibrary"):
            sys.modules.pop(mod, None)


def test_prefab_helpers_against_reference_capture(tmp_path):
    """drawComponent / eraseComponent / fabricate / apply_transform: a design using them must compile to the same
    files under the own API as under the reference (captured in tests/golden/scenes/prefabs)."""
    run(os.path.join(DESIGNS, "prefabs.py"), tmp_path)
    for f in FILES:
        assert (tmp_path / f).read_bytes() == open(os.path.join(SCENES, "prefabs", f), "rb").read(), f


def test_generated_design_corpus_matches_reference_digests(tmp_path):
    """72 generated designs (tests/designs/random_tree.py, reference-compatible mode: every helper of the
    reference's DesignCSG.py that shapes the build procedure) compiled by the own front-end must be byte-identical to
    what the reference Python wrote for the same scripts -- held as sha256 digests captured by
    tests/golden/make_frontend_corpus.py (scenecompiler.py:353-387,466-582; DesignCSG.py:45-203)."""
    sys.path.insert(0, DESIGNS)
    try:
        import random_tree
    finally:
        sys.path.pop(0)
    from designcsg_amd.run import run_design
    corpus = json.load(open(os.path.join(ROOT, "tests", "golden", "frontend_corpus.json")))["designs"]
    assert len(corpus) >= 50
    files = ["scene.txt", "buildprocedure.txt", "scene.cl", "exportConfig.txt", "arbitrary_data.hex"]
    import contextlib
    import io
    for e in corpus:
        text = random_tree.make_script(e["seed"], big=e["big"], smooth=False)
        assert hashlib.sha256(text.encode()).hexdigest() == e["script_sha256"], "generator drifted: regenerate the corpus"
        wd = tmp_path / ("s%d" % e["seed"])
        wd.mkdir()
        script = wd / "design.py"
        script.write_text(text)
        with contextlib.redirect_stdout(io.StringIO()):
            run_design(str(script), str(wd))
        for f in files:
            got = hashlib.sha256((wd / f).read_bytes()).hexdigest()
            assert got == e["sha256"][f], "seed %d: %s differs from the reference's output" % (e["seed"], f)
        assert sum(1 for _ in open(wd / "scene.txt")) == e["objects"]
