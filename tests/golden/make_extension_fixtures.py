#!/usr/bin/env python3
"""Scene fixtures of the smooth-join EXTENSION (build-procedure opcodes 6-8).

The reference has no such opcodes, so these scenes cannot be captured from it: they are compiled by this
package's own front-end (python -m designcsg_amd.run) from tests/designs/smooth_*.py and committed as data.
Parity for them is HIP path vs oracle only.  Runs anywhere (no /root/reference needed)."""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FILES = ["scene.txt", "buildprocedure.txt", "scene.cl", "exportConfig.txt"]
DESIGNS = ["smooth_blobs", "smooth256"]


def main():
    manifest = {}
    for name in DESIGNS:
        out_dir = os.path.join(HERE, "scenes", name)
        os.makedirs(out_dir, exist_ok=True)
        with tempfile.TemporaryDirectory() as scratch:
            subprocess.run([sys.executable, "-m", "designcsg_amd.run", os.path.join(ROOT, "tests", "designs", name + ".py"),
                            "--workdir", scratch], cwd=ROOT, check=True, stdout=subprocess.DEVNULL)
            for f in FILES:
                shutil.copyfile(os.path.join(scratch, f), os.path.join(out_dir, f))
        manifest[name] = {
            "script": "tests/designs/%s.py" % name, "compiled_by": "designcsg_amd.frontend (extension opcodes 6-8)",
            "objects": sum(1 for _ in open(os.path.join(out_dir, "scene.txt"))),
            "commands": len(open(os.path.join(out_dir, "buildprocedure.txt")).read().split("\n")),
            "sha256": {f: hashlib.sha256(open(os.path.join(out_dir, f), "rb").read()).hexdigest() for f in FILES}}
        print(name, manifest[name]["objects"], "objects", manifest[name]["commands"], "commands")
    with open(os.path.join(HERE, "scenes", "MANIFEST_extension.json"), "w") as fl:
        json.dump(manifest, fl, indent=1, sort_keys=True)
        fl.write("\n")


if __name__ == "__main__":
    main()
