"""`python -m designcsg_amd.run design.py [--export OUT_PREFIX] [--preview IMAGE.ppm] [--workdir DIR]
                                          [--devices 0,1,...] [--numerics relaxed] [--retopologize]`

Headless equivalent of the GUI's File>Run (+ File>Export): executes a design script against the own
design API (designcsg_amd/frontend on sys.path, so `from DesignCSG import *` resolves to it) in a
working directory, which leaves the compiled scene there (scene.txt, buildprocedure.txt, scene.cl,
arbitrary_data.hex, exportConfig.txt), and optionally exports <OUT_PREFIX>.stl/.ply on GPU 0 and/or renders the
640x480 preview the GUI's viewport shows (binary PPM; camera = the reference's start-up view, DrawPane.h:67).
--devices: several GPUs of this machine in ONE process (dcsg_export_multi: one Z slab and one host thread per GPU).
Opt-in, off by default: --numerics relaxed (faster arithmetic, within 1e-5, not bit-exact), --retopologize (T-junction
repair in the spirit of cms::retopologize).
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
    ap.add_argument("--devices", default=None, help="comma-separated GPU ids for a sharded export in this process")
    ap.add_argument("--numerics", default="exact", choices=["exact", "relaxed"])
    ap.add_argument("--retopologize", action="store_true")
    args = ap.parse_args()
    wd = run_design(args.design, args.workdir)
    devices = [int(v) for v in args.devices.split(",")] if args.devices else [args.device]
    if args.export or args.preview:
        from .evaluator import Evaluator, export_multi
        evs = []
        for dev in devices:
            e = Evaluator(dev)
            e.set_numerics(args.numerics)
            status, log = e.build_from_dir(wd)
            if status != 0:
                sys.exit("Error building export kernel:\n" + log)
            evs.append(e)
        ev = evs[0]
    if args.preview:
        img = ev.render_preview(campos=(0.0, 0.0, -args.zoom))
        with open(args.preview, "wb") as fl:
            fl.write(b"P6\n640 480\n255\n" + img.tobytes())
    if args.export and len(evs) > 1:
        if args.retopologize:
            sys.exit("--retopologize needs the whole mesh on one GPU")
        sts = export_multi(evs, open(os.path.join(wd, "exportConfig.txt")).read(), args.export + ".stl", args.export + ".ply")
        print("exported %d triangles on %d GPUs in %.3f s (per slab: %s)"
              % (sum(int(s.triangles) for s in sts), len(evs), max(s.t_total for s in sts), [int(s.triangles) for s in sts]))
    elif args.export:
        st = ev.export(open(os.path.join(wd, "exportConfig.txt")).read(), args.export + ".stl", args.export + ".ply",
                       retopologize=args.retopologize)
        print("exported %d triangles in %.3f s (bbox %.3f, sweep %.3f, mesh %.3f, projection %.3f, write %.3f)"
              % (st.triangles, st.t_total, st.t_bbox, st.t_sweep, st.t_mesh, st.t_project,
                 st.t_download + st.t_write_stl + st.t_write_ply))
    if args.export or args.preview:
        for e in evs:
            e.close()


if __name__ == "__main__":
    main()
