#!/usr/bin/env python3
"""Capture golden scene files by running design scripts on the REFERENCE Python.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box).  For every design it spawns a fresh interpreter (the reference
scene compiler is a process-wide singleton, scenecompiler.py:590-592) with
PYTHONPATH=/root/reference/master in a scratch directory, then copies the
files the reference wrote (scenecompiler.py:524,543,565 and
DesignCSG.py:221-237) into tests/golden/scenes/<name>/.

arbitrary_data.hex (524 288 bytes, scenecompiler.py:567-580) is all zeros for
every design here, so only its sha256 and non-zero count go into the manifest.

The outputs are DATA: the compiled scene format that is the input of the hot
path.  No reference .py file is copied; Design1/Design2 are executed from
where they lie under /root/reference.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/master"
OWN_DESIGNS = os.path.join(os.path.dirname(HERE), "designs")

DESIGNS = {
    # name -> path of the design script that is executed
    "design1": os.path.join(REF, "Designs", "Design1.py"),
    "design2": os.path.join(REF, "Designs", "Design2.py"),
    "sphere": os.path.join(OWN_DESIGNS, "sphere.py"),
    "mixed_groups": os.path.join(OWN_DESIGNS, "mixed_groups.py"),
    "flat256": os.path.join(OWN_DESIGNS, "flat256.py"),
    "grouped256": os.path.join(OWN_DESIGNS, "grouped256.py"),
    "ad_table": os.path.join(OWN_DESIGNS, "ad_table.py"),
    "prefabs": os.path.join(OWN_DESIGNS, "prefabs.py"),
    "const_table": os.path.join(OWN_DESIGNS, "const_table.py"),
    "global_state": os.path.join(OWN_DESIGNS, "global_state.py"),
    "userbrush128": os.path.join(OWN_DESIGNS, "userbrush128.py"),
}
FILES = ["scene.txt", "buildprocedure.txt", "scene.cl", "exportConfig.txt"]


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures can only be regenerated in the build container")
    manifest = {}
    for name, script in DESIGNS.items():
        out_dir = os.path.join(HERE, "scenes", name)
        os.makedirs(out_dir, exist_ok=True)
        with tempfile.TemporaryDirectory() as scratch:
            env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
            subprocess.run([sys.executable, script], cwd=scratch, env=env, check=True,
                           stdout=subprocess.DEVNULL)
            for f in FILES:
                shutil.copyfile(os.path.join(scratch, f), os.path.join(out_dir, f))
            ad = open(os.path.join(scratch, "arbitrary_data.hex"), "rb").read()
            nz = np.flatnonzero(np.frombuffer(ad, dtype="<f4"))
            if len(nz):  # only the used prefix is kept; the rest of the 131072 floats is zero
                with open(os.path.join(out_dir, "arbitrary_data.prefix.hex"), "wb") as fl:
                    fl.write(ad[:4 * (int(nz[-1]) + 1)])
            n_obj = sum(1 for _ in open(os.path.join(out_dir, "scene.txt")))
            n_cmd = len(open(os.path.join(out_dir, "buildprocedure.txt")).read().split("\n"))
            manifest[name] = {
                "script": os.path.relpath(script, "/root") if script.startswith(REF) else
                          os.path.relpath(script, os.path.dirname(HERE)),
                "objects": n_obj,
                "commands": n_cmd,
                "arbitrary_data_bytes": len(ad),
                "arbitrary_data_nonzero": int(np.count_nonzero(np.frombuffer(ad, dtype="<f4"))),
                "arbitrary_data_sha256": hashlib.sha256(ad).hexdigest(),
                "sha256": {f: hashlib.sha256(open(os.path.join(out_dir, f), "rb").read()).hexdigest()
                           for f in FILES},
            }
            print(name, n_obj, "objects", n_cmd, "commands")
    with open(os.path.join(HERE, "scenes", "MANIFEST.json"), "w") as fl:
        json.dump(manifest, fl, indent=1, sort_keys=True)
        fl.write("\n")


if __name__ == "__main__":
    main()
