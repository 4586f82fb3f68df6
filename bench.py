#!/usr/bin/env python3
"""Benchmark of the hot path: the fused SDF + central-difference-normal lattice sweep.

Metric (BASELINE.json): Mvoxels/s of SDF+normal evaluation.  A step is one sweep of the whole
(res+1)^3 lattice of the workload into buffers resident in HBM; voxels = res^3.  Default workload
is BASELINE config 2 (single sphere, 512^3, box from the reference's own stage-A estimate).

N > 1 (one process per GPU, torch.distributed over RCCL), default `--scaling strong` on the 1024^3 sphere lattice (the
other size BASELINE.json's metric is quoted on): ONE fixed lattice cut into N Z slabs; every rank
sweeps its slab and receives the boundary plane it shares with its upper neighbour over RCCL send/recv -- the one plane a
slab's mesher needs beyond its own, and the only exchange the path has.  Total work is fixed, value = res^3 voxels /
time (max over ranks); rank 0 also times the same lattice on its own GPU afterwards (`single_gpu_same_workload`), so
that the line carries the speed-up for the SAME workload.  `--scaling weak` stacks one workload-sized box per rank along
z instead (per-GPU work fixed, value = N * res^3 / time).  `--workload design2_1024` is BASELINE config 4's scene
(compute-bound: 75 ms per sweep on one GPU).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
# Ceiling for VALU issue: the vendor's FP32 vector peak (MI355X_MICROARCH.md: 157.3 TFLOP/s = 256 CUs x 4 SIMD-32 x one
# wave64 v_fma_f32 per 2 cycles x 2.4 GHz) as wave64 instructions per second.  Rounds 1-2 divided by a MEASURED v_fma rate
# (8.3e11/s), which a real kernel then exceeded (Design2: "fraction" 1.18): the chip holds different clocks under
# different instruction mixes, so only the fixed figure is a ceiling.
VALU_PEAK_WAVE_INSTR_PER_S = 157.3e12 / 2 / 64
BYTES_PER_POINT = 16   # 4 B SDF + 12 B normal written, nothing read

WORKLOADS = {
    # name: (fixture scene, grid level, box centre, box diameter)  -- boxes = reference stage-A goldens
    "sphere512": ("sphere", 9, [0.0, 0.0, 0.0], 5.3515625),
    "sphere1024": ("sphere", 10, [0.0, 0.0, 0.0], 5.3515625),
    "design2_512": ("design2", 9, [0.0, -0.0390625, 0.0], 9.3359375),
    "design2_1024": ("design2", 10, [0.0, -0.0390625, 0.0], 9.3359375),
    "flat256_512": ("flat256", 9, [0.0, 0.0, 0.0], 10.0),
    "flat256_1024": ("flat256", 10, [0.0, 0.0, 0.0], 10.0),
    "grouped256_512": ("grouped256", 9, [0.0, 0.0, 0.0], 10.0),
    "userbrush128_512": ("userbrush128", 9, [0.0, 0.0, 0.0], 10.0),  # 128 evaluations of the design's own brushes: threaded per-scene code
    "smooth256_512": ("smooth256", 9, [0.0, 0.0, 0.0], 10.0),     # smooth-join extension (no culling)
    "smooth256_1024": ("smooth256", 10, [0.0, 0.0, 0.0], 10.0),
}


def host_cores():
    """Threads this job may actually use: the cgroup CPU quota if there is one, else the affinity
    mask, capped at the GPU box's per-GPU CPU share (16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, int(os.environ.get("DCSG_CPU_THREADS", "16")))


def cpu_baseline(scene_dir, center, diameter, res, budget_s=15.0):
    """The oracle (own CPU restatement, kind "port") timed on this host's cores on a bounded sample
    of the same workload: whole lattice planes around the middle of the box."""
    from oracle import oracle as O
    O.build()
    osc = O.Scene.from_dir(scene_dir)
    cores = host_cores()
    n1 = res + 1
    z = n1 // 2
    osc.eval_lattice(center, diameter, res, z, z + 1, threads=cores)  # warm up the thread pool, untimed
    t0 = time.perf_counter()
    osc.eval_lattice(center, diameter, res, z, z + 2, threads=cores)
    t1 = (time.perf_counter() - t0) / 2
    want = max(1, int(budget_s / max(t1, 1e-5)))  # planes worth ~budget_s of CPU time
    chunk = min(64, n1)                            # planes per call: bounds the result buffers (64*n1^2*16 B)
    pts, planes, z = 0, 0, 0
    t0 = time.perf_counter()
    while planes < want:
        k = min(chunk, want - planes, n1 - z)
        osc.eval_lattice(center, diameter, res, z, z + k, threads=cores)
        planes += k
        pts += k * n1 * n1
        z = (z + k) % n1                           # wraps: several passes over the lattice if it is cheap
    dt = time.perf_counter() - t0
    # voxels per lattice point, so that the unit matches `value`
    mvox = pts * (res ** 3 / n1 ** 3) / dt / 1e6
    return {"value": round(mvox, 3), "unit": "Mvoxels/s", "cores": cores, "kind": "port",
            "sample": "oracle fused SDF+normal on %d lattice planes of %d^2 points (%.2f passes over the %d^3 lattice, "
                      "%d points, %.1f s)" % (planes, n1, planes / n1, n1, pts, dt)}


def reference_kernel_same_box(scene, scene_dir, center, diameter, res):
    """The REFERENCE's own kernel k2 (k2.cl + the scene's brush text, compiled for gfx950 by `make -C oracle ref-k2`
    with the reference's build options) on this box's GPU, launched as Evaluator.cpp:149-152,195-198 launches it:
    one batch of MAX_EVAL_POINTS = 16 777 216 lattice points, one SDF launch and one normal launch, LOCAL SIZE 1.
    Part of the cpu_baseline leg (checker code, oracle/ref_k2.py); None where the code objects were not built."""
    from oracle import ref_k2 as R
    if not os.path.exists(R.hsaco_path(scene, "default")):
        return None
    import numpy as np
    n1 = res + 1
    n = min(R.MAX_EVAL_POINTS, n1 ** 3)
    idx = np.arange(n, dtype=np.int64)
    d, w = np.float32(diameter), np.float32(res)
    axis = lambda c, i: (np.float32(c) - d * np.float32(0.5)) + (d * i.astype(np.float32)) / w     # ISV.hpp:103-109
    pts = np.stack([axis(center[0], idx % n1), axis(center[1], (idx // n1) % n1), axis(center[2], idx // (n1 * n1))], -1)
    k = R.RefK2(scene_dir, "default", scene_name=scene, max_points=n)
    try:
        k.eval_sdf_at_points(pts[:1 << 16])            # untimed: first launch of the module
        t0 = time.perf_counter()
        k.eval_sdf_at_points(pts)
        ms_sdf = k.last_kernel_ms
        k.eval_normal_at_points(pts)
        ms_nrm = k.last_kernel_ms
        wall = time.perf_counter() - t0
    finally:
        k.close()
    vox = n * (res ** 3 / n1 ** 3)
    return {"value": round(vox / ((ms_sdf + ms_nrm) * 1e-3) / 1e6, 2), "unit": "Mvoxels/s", "kind": "reference",
            "kernel_ms": {"sdf": round(ms_sdf, 3), "normal": round(ms_nrm, 3)},
            "with_blocking_transfers_Mvoxels_per_s": round(vox / wall / 1e6, 2),
            "sample": "reference kernel k2 (oracle/_ref/k2_%s_default.hsaco: the reference's text and build options, gfx950) on "
                      "the first %d points of the %d^3 lattice, one SDF + one normal launch at local size 1 "
                      "(Evaluator.cpp:149-152,195-198); value = kernel time only (HIP events), the second figure includes "
                      "the reference's three blocking transfers per call" % (scene, n, n1)}


# kernel that a fused SDF+normal sweep launches, by interpreter variant (dcsg_api.cpp: sweep_device); the SDF-only
# sweep launches the `_d` instantiation of the same kernel
KERNEL_VARIANT = {0: "dcsg_sweep_f", 1: "dcsg_sweep_s4_f", 2: "dcsg_sweep_s8_f", 3: "dcsg_sweep_sp_f", 4: "dcsg_sweep_c4_f"}
SDF_ONLY_KERNEL = {0: "dcsg_sweep_d", 1: "dcsg_sweep_s4_d", 2: "dcsg_sweep_s8_d", 3: "dcsg_sweep_sp_d", 4: "dcsg_sweep_c4_d"}


def time_sweeps(ev, center, diameter, res, reps=10, settle_s=0.15):
    """Device time (HIP events on the launch stream) of the fused and of the SDF-only sweep of the whole lattice with the
    scene currently loaded in `ev`.  Secondary lines only: outside the timed region of the headline number."""
    import ctypes as C
    import torch
    from designcsg_amd import _lib
    L, h = _lib.lib(), ev._h
    n1 = res + 1
    sdf = torch.empty(n1 ** 3, dtype=torch.float32, device="cuda")
    nrm = torch.empty(3 * n1 ** 3, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()
    c3, d3, sp = (C.c_float * 3)(*center), (C.c_float * 3)(diameter, diameter, diameter), C.c_void_p(stream.cuda_stream)
    out = {}
    for key, nptr in (("fused", nrm.data_ptr()), ("sdf_only", None)):
        launch = lambda: L.dcsg_sweep_lattice_device(h, c3, d3, res, 0, n1, C.c_void_p(sdf.data_ptr()), C.c_void_p(nptr) if nptr else None, sp)
        # untimed lead-in until the clocks have settled (see --settle-s): settle_s seconds of launches, then two more
        t_end = time.perf_counter() + settle_s
        while time.perf_counter() < t_end:
            for _ in range(8):
                rc = launch()
                if rc:
                    _lib.check(h, rc)
                if time.perf_counter() > t_end + 0.5:    # sweeps of tens of milliseconds (Design2 at 1024^3)
                    break
            torch.cuda.synchronize()
        reg = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for i in range(reps + 2):
            if i == 2:
                reg[0].record(stream)
            rc = launch()
            if rc:
                _lib.check(h, rc)
        reg[1].record(stream)
        torch.cuda.synchronize()
        ms = reg[0].elapsed_time(reg[1]) / reps   # device time per launch: events around the region of `reps` launches
        bytes_ = (16 if nptr else 4) * n1 ** 3
        out[key] = {"kernel_ms": round(ms, 4), "GB_per_s": round(bytes_ / (ms * 1e-3) / 1e9, 1),
                    "frac_of_hbm_peak": round(bytes_ / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "Mvoxels_per_s": round(res ** 3 / (ms * 1e-3) / 1e6, 1)}
    del sdf, nrm
    torch.cuda.empty_cache()
    return out


def secondary_sweeps(scene_dir, device, center, diameter, res):
    """N=1 only, outside the timed region: the same workload through (a) the LDS-table stack interpreter north_star
    describes (`_s4` kernels: op / object tables staged in LDS, no per-scene code), (b) the opt-in relaxed-numerics
    module (NOT bit-exact; tests/test_gpu_relaxed.py bounds it), and (c) the exact default at 1024^3, the other lattice
    size BASELINE.json's metric is quoted on."""
    from designcsg_amd.evaluator import Evaluator
    out = {}
    os.environ["DCSG_SPECIALISE"] = "0"
    try:
        ev = Evaluator(device)
        assert ev.build_from_dir(scene_dir)[0] == 0
        k = ev.scene_info()["interpreter"]
        out["lds_table_interpreter"] = dict(time_sweeps(ev, center, diameter, res), kernel=KERNEL_VARIANT[k], numerics="exact")
        ev.close()
    finally:
        os.environ.pop("DCSG_SPECIALISE", None)
    ev = Evaluator(device)
    ev.set_numerics("relaxed")
    assert ev.build_from_dir(scene_dir)[0] == 0
    out["relaxed_numerics"] = dict(time_sweeps(ev, center, diameter, res), kernel=KERNEL_VARIANT[ev.scene_info()["interpreter"]],
                                   numerics="relaxed (opt-in: hardware sqrt/rsq, FMA contraction; within 1e-5, not bit-exact)")
    out["relaxed_numerics_1024"] = dict(time_sweeps(ev, center, diameter, 2 * res, reps=5), numerics="relaxed", lattice_points=(2 * res + 1) ** 3)
    ev.close()
    ev = Evaluator(device)
    assert ev.build_from_dir(scene_dir)[0] == 0
    out["exact_1024"] = dict(time_sweeps(ev, center, diameter, 2 * res, reps=5), kernel=KERNEL_VARIANT[ev.scene_info()["interpreter"]],
                             numerics="exact", lattice_points=(2 * res + 1) ** 3)
    ev.close()
    return out


def export_wall_clock(ev=None):
    """Second half of BASELINE.json's metric: export wall-clock seconds, scene files on disk -> .stl and .ply closed, for
    Design2 at 512^3 and 1024^3 (BASELINE configs 3 and 4) on this one GPU, as SURVEY.md 8(d) defines it: each figure comes
    from a FRESH child process (tools/export_from_disk.py) that creates the context, loads the scene from its files and
    exports -- once with the JIT cache warm and once with DCSG_JIT_NO_CACHE=1 (the brush text compiled from scratch).
    Outside the timed region of the headline number."""
    import subprocess
    out = []
    # Design2 at 512^3 and 1024^3 (BASELINE configs 3, 4), then config 5's two scenes at its stated size
    for scene, levels in (("design2", (6, 8, 9)), ("design2", (7, 9, 10)), ("flat256", (7, 9, 10)), ("grouped256", (7, 9, 10))):
        d = os.path.join(ROOT, "tests", "golden", "scenes", scene)
        entry = {"config": "%s levels %d/%d/%d (%d^3 voxels)" % (("Design2" if scene == "design2" else scene,) + levels + (1 << levels[2],))}
        entry["export_roofline"] = export_roofline(scene, levels[2])
        for key, extra in ((("warm_jit_cache", []), ("cold_jit", ["--cold-jit"])) if scene == "design2" else (("fresh_process", []),)):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "export_from_disk.py"), "--scene-dir", d, "--levels"]
                               + [str(v) for v in levels] + extra, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                entry[key] = {"error": (r.stdout + r.stderr)[-400:]}
                continue
            j = json.loads(r.stdout.strip().split("\n")[-1])
            entry.setdefault("triangles", j["triangles"])
            entry.setdefault("stl_bytes", j["stl_bytes"])
            entry.setdefault("ply_bytes", j["ply_bytes"])
            entry[key] = {k: j[k] for k in ("disk_to_files_s", "with_context_s", "create_context_s", "read_files_and_load_scene_s",
                                            "first_export_s", "steady_export_s", "load_scene_ms")}
        entry["seconds"] = entry.get("warm_jit_cache", entry.get("fresh_process", {})).get("disk_to_files_s")
        out.append(entry)
    return out


def profiled_traffic(kernel, workload, numerics="exact"):
    """HBM bytes and VALU instructions per SWEEP (= per launch unless the lattice needs several launches) of `kernel` from the committed PMC summaries (profiles/<tag>_pmc_summary.json:
    separate rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of this same command, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950), taken from the newest profile of the same workload."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        tag = os.path.basename(f)[:-len("_pmc_summary.json")]
        try:
            b = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench.json")))
            pmc = json.load(open(f))
        except (OSError, ValueError):
            continue
        if b.get("config", {}).get("workload") == workload and b.get("config", {}).get("numerics", "exact") == numerics \
                and kernel in pmc and "hbm_bytes_per_launch" in pmc[kernel]:
            lps = pmc[kernel].get("launches_per_sweep", 1)
            valu = pmc[kernel]["counters_mean_per_launch"].get("SQ_INSTS_VALU")
            best = (pmc[kernel].get("hbm_bytes_per_sweep", pmc[kernel]["hbm_bytes_per_launch"])["total"], "profiles/" + os.path.basename(f),
                    valu * lps if valu else None)
    return best


def profiled_valu_per_point(kernel, scene):
    """VALU wave64 instructions per lattice point of `kernel` on `scene`, from the newest committed PMC summary of a bench
    run on that scene (profiles/<tag>_pmc_summary.json + <tag>_bench.json); None when there is none."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        tag = os.path.basename(f)[:-len("_pmc_summary.json")]
        try:
            b = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench.json")))
            pmc = json.load(open(f))
        except (OSError, ValueError):
            continue
        cfg = b.get("config", {})
        if cfg.get("workload", "").startswith(scene + " scene") and cfg.get("numerics", "exact") == "exact" and kernel in pmc:
            v = pmc[kernel]["counters_mean_per_launch"].get("SQ_INSTS_VALU")
            pts = cfg.get("lattice_points")
            if v and pts:
                # counters are means per LAUNCH; a sweep of more than 0xf0000000 points is several launches
                best = (v * max(1, cfg.get("launches_per_sweep", 1)) * 64.0 / pts, "profiles/" + os.path.basename(f))
    return best


def valu_issue(kernel_ms, lattice_points, valu_per_point, source):
    rate = valu_per_point * lattice_points / 64.0 / (kernel_ms * 1e-3)
    return {"valu_per_lattice_point": round(valu_per_point, 1), "achieved_wave_instr_per_s": round(rate, 0),
            "peak_wave_instr_per_s": round(VALU_PEAK_WAVE_INSTR_PER_S, 0), "peak_source": "vendor FP32 vector peak 157.3 TFLOP/s / 128",
            "frac": round(rate / VALU_PEAK_WAVE_INSTR_PER_S, 4), "instruction_count_source": source}


def design2_sweeps(device):
    """N=1 only, outside the timed region: the scene BASELINE.json's export target names (Design2: one evaluation = 63 boxes
    + a disc, arithmetic-bound) swept densely at 512^3 and 1024^3 -- kernel time, fraction of the HBM peak, and VALU issue
    rate against the fixed ceiling (instruction counts per lattice point from the committed PMC profile of this scene)."""
    from designcsg_amd.evaluator import Evaluator
    scene, _, center, diameter = WORKLOADS["design2_512"]
    ev = Evaluator(device)
    assert ev.build_from_dir(os.path.join(ROOT, "tests", "golden", "scenes", scene))[0] == 0
    out = {}
    for key, res, reps in (("design2_512", 512, 5), ("design2_1024", 1024, 3)):
        t = time_sweeps(ev, center, diameter, res, reps=reps, settle_s=0.1)
        n1 = res + 1
        for form, kern in (("fused", "dcsg_sweep_sp_f"), ("sdf_only", "dcsg_sweep_sp_d")):
            t[form]["kernel"] = kern
            p = profiled_valu_per_point(kern, scene)
            if p:
                t[form]["valu_issue"] = valu_issue(t[form]["kernel_ms"], n1 ** 3, p[0], p[1])
        out[key] = dict(t, lattice_points=n1 ** 3, numerics="exact")
    ev.close()
    return out


def many_object_sweeps(device):
    """N=1 only, outside the timed region: BASELINE config 5's scenes (256 standard brushes: flat and grouped; the reference
    cannot even load them, DrawPane.h:14-15) and a 128-brush scene of the design's own OpenCL brushes, swept densely at
    512^3 -- kernel time of the fused and of the SDF-only sweep, fraction of the HBM peak, and the VALU issue rate against
    the fixed ceiling (instruction counts per lattice point from the committed PMC profile of the same workload)."""
    from designcsg_amd.evaluator import Evaluator
    out = {}
    for key in ("flat256_512", "grouped256_512", "userbrush128_512"):
        scene, level, center, diameter = WORKLOADS[key]
        ev = Evaluator(device)
        assert ev.build_from_dir(os.path.join(ROOT, "tests", "golden", "scenes", scene))[0] == 0
        k = ev.scene_info()["interpreter"]
        res = 1 << level
        t = time_sweeps(ev, center, diameter, res, reps=3, settle_s=0.05)
        for form, kern in (("fused", KERNEL_VARIANT[k]), ("sdf_only", SDF_ONLY_KERNEL[k])):
            t[form]["kernel"] = kern
            p = profiled_valu_per_point(kern, scene)
            if p:
                t[form]["valu_issue"] = valu_issue(t[form]["kernel_ms"], (res + 1) ** 3, p[0], p[1])
        out[key] = dict(t, lattice_points=(res + 1) ** 3, numerics="exact", objects=ev.scene_info()["objects"])
        ev.close()
    return out


def export_roofline(scene, grid_level):
    """Where an export's GPU time goes, from the newest committed rocprofv3 summary of that export
    (profiles/r*_export_<scene>_<size>.json, written by profiles/summarise_export.py from kernel-trace and PMC passes of
    tools/bench_export.py): the dominant kernel with its share, its VALU issue fraction and its HBM fraction."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_export_%s_%d.json" % (scene, 1 << grid_level))))
    if not files:
        return None
    j = json.load(open(files[-1]))
    k = j["dominant_kernel"]
    o = j["kernels"][k]
    top = sorted(j["kernels"].items(), key=lambda kv: -kv[1]["ms_per_export_settled_median"])[:4]
    return {"source": "profiles/" + os.path.basename(files[-1]), "gpu_ms_per_export_settled": j["gpu_ms_per_export_settled"],
            "dominant_kernel": k, "share_of_export_gpu_time": o["share_of_export_gpu_time"], "kernel_ms": o["ms_per_export_settled_median"],
            "bound": "valu", "frac": o.get("valu_frac_of_ceiling"), "valu_per_SIMD_cycle": o.get("valu_per_SIMD_cycle"),
            "hbm_frac": o.get("hbm_frac_of_peak"),
            "kernels": {n: {"ms": v["ms_per_export_settled_median"], "valu_frac": v.get("valu_frac_of_ceiling"), "hbm_frac": v.get("hbm_frac_of_peak")}
                        for n, v in top}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle-s", type=float, default=0.3,
                    help="seconds of untimed steps BEFORE the --warmup steps, so that the GPU has left its idle clock state "
                         "(a 513^3 sweep takes 0.57 ms on a GPU that was idle 20 ms ago and 0.44 ms once its clocks have "
                         "settled; W = 5 steps are 3 ms).  Reported as config.settle_steps; 0 disables it")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: sphere512 on one GPU (BASELINE config 2), sphere1024 on several")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--numerics", default=os.environ.get("DCSG_NUMERICS", "exact"), choices=["exact", "relaxed"],
                    help="exact (default; bit-identical to the reference arithmetic) or the opt-in relaxed module -- a relaxed "
                         "line says so in config.numerics and is not the headline number")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary sweep lines (LDS-table interpreter, "
                    "relaxed numerics, 1024^3)")
    ap.add_argument("--no-export", action="store_true", help="skip the export wall-clock lines (second half of the metric)")
    ap.add_argument("--verify-halo", dest="verify_halo", action="store_true", default=None,
                    help="after the timed loop (outside the timed region), check the received halo plane against this rank's "
                         "own evaluation of it; default: on for N > 1")
    ap.add_argument("--no-verify-halo", dest="verify_halo", action="store_false")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N>1: strong = the single workload lattice split into N Z slabs (default); "
                         "weak = one workload-sized lattice box per rank, stacked along z")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default, what the driver runs); gloo = rehearsal of the N>1 logic on "
                         "a box with fewer GPUs than ranks: halo planes are staged through host memory")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from designcsg_amd.evaluator import Evaluator

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                 % (args.gpus, world, args.gpus))
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())  # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    if args.verify_halo is None:
        args.verify_halo = world > 1
    # Proof that the communicator really spans N ranks: a sum of ones over the process group, on the
    # device for nccl (= RCCL), reported as config.rccl_ranks next to the backend torch.distributed says it runs
    comm = {"backend": None, "ranks_seen": 1}
    if world > 1:
        ones = torch.ones(1, dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        comm = {"backend": dist.get_backend(), "ranks_seen": int(ones.item())}
        if comm["ranks_seen"] != world:
            sys.exit("all_reduce over the %s group counted %d ranks, WORLD_SIZE is %d" % (comm["backend"], comm["ranks_seen"], world))

    if args.workload is None:
        args.workload = "sphere512" if world == 1 else "sphere1024"
    scene, grid_level, center, diameter = WORKLOADS[args.workload]
    scene_dir = os.path.join(ROOT, "tests", "golden", "scenes", scene)
    res = 1 << grid_level
    n1 = res + 1
    ev = Evaluator(local_rank)
    ev.set_numerics(args.numerics)
    status, log = ev.build_from_dir(scene_dir)
    if status != 0:
        sys.exit("scene build failed:\n" + log)

    # Z slab of this rank: planes [lo, hi) are its own, plane hi is the halo its mesher reads (the upper
    # neighbour's first plane); the last rank owns plane hi as well.
    weak = args.scaling == "weak" or world == 1
    if weak:
        lo, hi = 0, res                                   # a whole box per rank ...
        center = [center[0], center[1], center[2] + rank * diameter]   # ... stacked along z
        total_voxels = world * res ** 3
    else:
        lo, hi = rank * res // world, (rank + 1) * res // world
        total_voxels = res ** 3
    own_hi = hi if rank < world - 1 else hi + 1
    planes = hi - lo + 1
    plane = n1 * n1
    sdf = torch.empty(planes * plane, dtype=torch.float32, device="cuda")
    nrm = torch.empty(planes * plane * 3, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()

    halo_sdf, halo_nrm = sdf[(hi - lo) * plane:], nrm[(hi - lo) * plane * 3:]  # plane hi (neighbour's first)

    # Pre-bound C call: the per-step host work is two launches and one grouped RCCL exchange.
    import ctypes as C
    from designcsg_amd import _lib
    L, h = _lib.lib(), ev._h
    c3 = (C.c_float * 3)(*center)
    d3 = (C.c_float * 3)(diameter, diameter, diameter)
    sp = C.c_void_p(stream.cuda_stream)

    def sweep(z_a, z_b):
        off = z_a - lo
        rc = L.dcsg_sweep_lattice_device(h, c3, d3, res, z_a, z_b, C.c_void_p(sdf.data_ptr() + off * plane * 4),
                                         C.c_void_p(nrm.data_ptr() + off * plane * 12), sp)
        if rc:
            _lib.check(h, rc)

    send_first = world > 1 and rank > 0
    recv_halo = world > 1 and rank < world - 1
    if args.backend == "nccl" and world > 1:
        ops = []
        if send_first:
            ops += [dist.P2POp(dist.isend, sdf[:plane], rank - 1), dist.P2POp(dist.isend, nrm[:3 * plane], rank - 1)]
        if recv_halo:
            ops += [dist.P2POp(dist.irecv, halo_sdf, rank + 1), dist.P2POp(dist.irecv, halo_nrm, rank + 1)]

    def step():
        # own planes; the first one is what the lower neighbour needs, so it goes first and its transfer
        # (one grouped send/recv with both Z neighbours) overlaps the sweep of the rest of the slab
        first, staged, reqs = lo, None, []
        if send_first:
            sweep(lo, lo + 1)
            first = lo + 1
        if world > 1:
            if args.backend == "nccl":
                reqs = dist.batch_isend_irecv(ops) if ops else []
            else:
                if send_first:
                    reqs += [dist.isend(sdf[:plane].cpu(), rank - 1), dist.isend(nrm[:3 * plane].cpu(), rank - 1)]
                if recv_halo:
                    staged = (torch.empty(plane, dtype=torch.float32), torch.empty(3 * plane, dtype=torch.float32))
                    reqs += [dist.irecv(staged[0], rank + 1), dist.irecv(staged[1], rank + 1)]
        sweep(first, own_hi)
        for r in reqs:
            r.wait()
        if staged is not None:
            halo_sdf.copy_(staged[0])
            halo_nrm.copy_(staged[1])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Clock settling (untimed, declared in the output line): the GPU comes out of scene loading idle and needs tens of
    # milliseconds of work to reach its sustained clocks -- longer than W warm-up steps of a sub-millisecond kernel last.
    # One process: steps until --settle-s have passed.  Several ranks: a fixed number of steps on every rank (a step holds
    # a send / receive pair, so all ranks must run the same count).
    settle_steps = 0
    from_idle = None
    if args.settle_s > 0:
        if world == 1:
            # for the record: the same W + K procedure on the GPU as scene loading left it (idle clocks), reported next to
            # the headline as config.from_idle
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            t_i = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt_i = time.perf_counter() - t_i
            from_idle = {"ms_per_step": round(dt_i / args.steps * 1e3, 4),
                         "Mvoxels_per_s": round(total_voxels * args.steps / dt_i / 1e6, 2)}
            t_end = time.perf_counter() + args.settle_s
            while time.perf_counter() < t_end:
                for _ in range(8):
                    step()
                torch.cuda.synchronize()
                settle_steps += 8
        else:
            settle_steps = 64
            for _ in range(settle_steps):
                step()
    for _ in range(args.warmup):
        step()
    barrier()
    # kernel time, live: device events on the stream the sweep is launched on
    k_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    region[0].record(stream)
    for i in range(args.steps):
        if world > 1:
            k_ev[i][0].record(stream)
        step()
        if world > 1:
            k_ev[i][1].record(stream)
    region[1].record(stream)
    barrier()
    dt = time.perf_counter() - t0
    halo_ok = None
    if args.verify_halo and world > 1:
        # the plane received from the upper neighbour must equal what this rank computes for it
        bad = torch.zeros(1, dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        if rank < world - 1:
            chk = torch.empty(plane, dtype=torch.float32, device="cuda")
            chn = torch.empty(3 * plane, dtype=torch.float32, device="cuda")
            ev.sweep_lattice_device(center, diameter, res, hi, hi + 1, chk.data_ptr(), chn.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            if not (torch.equal(chk.view(torch.int32), halo_sdf.view(torch.int32))
                    and torch.equal(chn.view(torch.int32), halo_nrm.view(torch.int32))):
                sys.stderr.write("rank %d: halo plane %d differs from the local evaluation\n" % (rank, hi))
                bad += 1
            del chk, chn
        dist.all_reduce(bad, op=dist.ReduceOp.SUM)   # every rank learns the verdict (no rank exits alone and hangs the rest)
        halo_ok = int(bad.item()) == 0
        if not halo_ok:
            dist.destroy_process_group()
            sys.exit("halo verification failed on %d rank(s)" % int(bad.item()))
    # Secondary line (N=1, outside the timed region): the SDF-only sweep of the same lattice -- the 4 B/point
    # kernel the export runs when normals are sampled on demand, and the one BASELINE.json's HBM target names.
    sdf_only = None
    if world == 1:
        reps = max(5, args.steps)
        lead = 2 + (64 if args.settle_s > 0 else 0)   # the timed loop above has just ended: a short lead-in is enough
        reg = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for i in range(reps + lead):
            if i == lead:
                reg[0].record(stream)
            rc = L.dcsg_sweep_lattice_device(h, c3, d3, res, lo, own_hi, C.c_void_p(sdf.data_ptr()), None, sp)
            if rc:
                _lib.check(h, rc)
        reg[1].record(stream)
        # ... and with an event pair around every launch (rounds 1-2's figure, 4-5 % higher on a 0.1 ms kernel: the
        # events' own turnaround), kept next to it
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record(stream)
            rc = L.dcsg_sweep_lattice_device(h, c3, d3, res, lo, own_hi, C.c_void_p(sdf.data_ptr()), None, sp)
            if rc:
                _lib.check(h, rc)
            b.record(stream)
        torch.cuda.synchronize()
        ms = reg[0].elapsed_time(reg[1]) / reps            # region / launches, as the headline kernel's time above
        ms_pairs = sum(a.elapsed_time(b) for a, b in evs) / reps
        gbs = 4 * (own_hi - lo) * plane / (ms * 1e-3) / 1e9
        sdf_only = {"kernel": SDF_ONLY_KERNEL.get(ev.scene_info()["interpreter"], KERNEL_VARIANT[ev.scene_info()["interpreter"]]),
                    "kernel_ms": round(ms, 4), "kernel_ms_method": "HIP events around a region of back-to-back launches / launches",
                    "kernel_ms_event_pair_per_launch": round(ms_pairs, 4), "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "Mvoxels_per_s": round(res ** 3 / (ms * 1e-3) / 1e6, 1),
                    "algorithmic_bytes_per_launch": 4 * (own_hi - lo) * plane}
    # N > 1, strong scaling: the same lattice on rank 0's GPU alone (outside the timed region; the other ranks wait)
    single = None
    if world > 1 and not weak:
        if rank == 0:
            single = time_sweeps(ev, center, diameter, res, reps=max(3, min(args.steps, 10)))["fused"]
        dist.barrier()
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        if args.backend == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        kernel = KERNEL_VARIANT[ev.scene_info()["interpreter"]]
        workload = ("%s scene, fused SDF+normal sweep of the %d^3 lattice (%d^3 voxels), box centre %s diameter %s"
                    % (scene, n1, res, center, diameter))
        ms_per_step = dt / args.steps * 1e3
        value = total_voxels * args.steps / dt / 1e6
        # The kernel's average launch duration from HIP events over the timed region: one event before the K steps, one
        # after, on the launch stream, nothing in between (N = 1: a step is exactly one launch of the kernel, so region / K
        # is what rocprofv3's average duration is compared with; it includes the gaps between launches).  N > 1: a step is
        # several launches plus the exchange, timed by an event pair per step.
        if world == 1:
            kernel_ms = region[0].elapsed_time(region[1]) / args.steps
        else:
            kernel_ms = sum(a.elapsed_time(b) for a, b in k_ev) / args.steps
        own_points = (own_hi - lo) * plane
        achieved = BYTES_PER_POINT * own_points / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "Mvoxels/s SDF+normal eval", "value": round(value, 2), "unit": "Mvoxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload,
                       "grid_level": grid_level, "numerics": args.numerics, "settle_steps": settle_steps, "from_idle": from_idle,
                       "parallelism": ("z-slab x%d: %d stacked %d^3-voxel boxes, boundary plane over RCCL" % (world, world, res)
                                       if weak else "z-slab x%d of one %d^3-voxel lattice, halo plane over RCCL" % (world, res)),
                       "lattice_points": (world * res + 1) * plane if weak else n1 ** 3,
                       "launches_per_sweep": -(-(own_hi - lo) // max(1, 0xf0000000 // plane)),
                       "comm_backend": comm["backend"], "rccl_ranks": comm["ranks_seen"] if comm["backend"] == "nccl" else None,
                       "ranks_seen_by_all_reduce": comm["ranks_seen"], "halo_verified": halo_ok},
            "from_idle": from_idle,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": kernel, "kernel_ms": round(kernel_ms, 4),
                         "kernel_ms_method": "HIP events around the timed region / K launches" if world == 1 else "event pair per step",
                         "algorithmic_bytes_per_launch": BYTES_PER_POINT * own_points},
        }
        if single:
            out["single_gpu_same_workload"] = dict(single, speedup=round(value / single["Mvoxels_per_s"], 3),
                                                   note="the same %d^3 lattice swept by rank 0's GPU alone (kernel time)" % n1)
        if sdf_only:
            out["sdf_only_sweep"] = sdf_only
        tr = profiled_traffic(kernel, workload, args.numerics) if world == 1 else None
        if tr:
            out["roofline"]["traffic"] = tr[0]
            out["roofline"]["traffic_source"] = tr[1]
            if tr[2]:
                # VALU issue: wave64 VALU instructions issued per launch (PMC, same profile) against
                # the fixed ceiling above
                out["roofline"]["valu_issue"] = valu_issue(kernel_ms, own_points, tr[2] * 64.0 / own_points, tr[1])
        if world == 1 and not args.no_secondary:
            out["secondary_sweeps"] = secondary_sweeps(scene_dir, local_rank, center, diameter, res)
            out["secondary_sweeps"].update(design2_sweeps(local_rank))
            out["secondary_sweeps"].update(many_object_sweeps(local_rank))
        if world == 1 and not args.no_export:
            out["export_wall_clock"] = export_wall_clock()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene_dir, center, diameter, res)
            ref = reference_kernel_same_box(scene, scene_dir, center, diameter, res)
            if ref:
                out["cpu_baseline"]["reference_kernel_same_box"] = ref
        print(json.dumps(out))
    ev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
