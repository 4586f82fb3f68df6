"""Same-box A/B of the lattice sweeps of two builds of this tree (diagnostic):
    python tools/ab_sweep.py TREE_A TREE_B [--scene sphere] [--level 9] [--rounds 3]
Each tree is a checkout with its own designcsg_amd/_build; every measurement runs in a child process (one library per
process), alternating A, B, A, B ... so clock / box drift shows up as spread inside a tree rather than between them."""
import json
import os
import subprocess
import sys

CHILD = r'''
import json, os, sys
tree, scene, level, numerics = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, tree)
os.chdir(tree)
import bench
from designcsg_amd.evaluator import Evaluator
d = os.path.join(tree, "tests", "golden", "scenes", scene)
ev = Evaluator(0)
ev.set_numerics(numerics)
st, log = ev.build_from_dir(d)
assert st == 0, log
cfg = open(os.path.join(d, "exportConfig.txt")).read()
half = float(cfg.split()[0])
center, diameter, _ = ev.estimate_bounding_box(half)
out = bench.time_sweeps(ev, center, diameter, 1 << level, reps=20)
print(json.dumps({k: v["kernel_ms"] for k, v in out.items()}))
'''


def main():
    a, b = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
    opt = lambda n, d: sys.argv[sys.argv.index(n) + 1] if n in sys.argv else d
    scene, level, rounds, numerics = opt("--scene", "sphere"), opt("--level", "9"), int(opt("--rounds", "3")), opt("--numerics", "exact")
    envs = {"A": opt("--env-a", ""), "B": opt("--env-b", "")}   # "KEY=VALUE,KEY2=VALUE2" for one side
    res = {"A": [], "B": []}
    for r in range(rounds):
        for name, t in (("A", a), ("B", b)):
            env = dict(os.environ)
            env.update(kv.split("=", 1) for kv in envs[name].split(",") if kv)
            p = subprocess.run([sys.executable, "-c", CHILD, t, scene, level, numerics], capture_output=True, text=True, env=env)
            if p.returncode:
                sys.exit(p.stderr[-2000:])
            res[name].append(json.loads(p.stdout.strip().splitlines()[-1]))
    for name, t in (("A", a), ("B", b)):
        for key in ("fused", "sdf_only"):
            v = [x[key] for x in res[name]]
            print("%s %-34s %-24s %-9s min %.4f  all %s" % (name, os.path.relpath(t), envs[name], key, min(v), v))


if __name__ == "__main__":
    main()
