"""`python -m designcsg_amd.run design.py [--export OUT_PREFIX] [--preview IMAGE.ppm] [--workdir DIR]`

Headless equivalent of the GUI's File>Run (+ File>Export): executes a design script against the own
design API (designcsg_amd/frontend on sys.path, so `from DesignCSG import *` resolves to it) in a
working directory, which leaves the compiled scene there (scene.txt, buildprocedure.txt, scene.cl,
arbitrary_data.hex, exportConfig.txt), and optionally exports <OUT_PREFIX>.stl/.ply on GPU 0 and/or renders the
640x480 preview the GUI's viewport shows (binary PPM; camera = the reference's start-up view, DrawPane.h:67).
"""
import argparse
import os
import runpy
import sys

FRONTEND = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frontend")


def run_design(script, workdir):
    """Execute one design script with the own API; returns the working directory."""
    script = os.path.abspath(script)
    os.makedirs(workdir, exist_ok=True)
    old_cwd, old_path = os.getcwd(), list(sys.path)
    # the scene compiler is a per-process singleton: start every run from fresh modules
    for m in ("DesignCSG", "scenecompiler", "designlibrary"):
        sys.modules.pop(m, None)
    sys.path.insert(0, FRONTEND)
    os.chdir(workdir)
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        os.chdir(old_cwd)
        sys.path[:] = old_path
        for m in ("DesignCSG", "scenecompiler", "designlibrary"):
            sys.modules.pop(m, None)
    return workdir


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("design")
    ap.add_argument("--workdir", default=".")
    ap.add_argument("--export", metavar="OUT_PREFIX")
    ap.add_argument("--preview", metavar="IMAGE.ppm")
    ap.add_argument("--zoom", type=float, default=15.0, help="camera distance for --preview")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    wd = run_design(args.design, args.workdir)
    if args.export or args.preview:
        from .evaluator import Evaluator
        ev = Evaluator(args.device)
        status, log = ev.build_from_dir(wd)
        if status != 0:
            sys.exit("Error building export kernel:\n" + log)
    if args.preview:
        img = ev.render_preview(campos=(0.0, 0.0, -args.zoom))
        with open(args.preview, "wb") as fl:
            fl.write(b"P6\n640 480\n255\n" + img.tobytes())
    if args.export:
        st = ev.export(open(os.path.join(wd, "exportConfig.txt")).read(), args.export + ".stl", args.export + ".ply")
        print("exported %d triangles in %.3f s (bbox %.3f, sweep %.3f, mesh %.3f, projection %.3f, write %.3f)"
              % (st.triangles, st.t_total, st.t_bbox, st.t_sweep, st.t_mesh, st.t_project,
                 st.t_download + st.t_write_stl + st.t_write_ply))
    if args.export or args.preview:
        ev.close()


if __name__ == "__main__":
    main()
