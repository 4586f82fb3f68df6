#!/bin/bash
# Per-launch durations of the mesher's kernels in one export (the last of three), from a rocprofv3 kernel trace.
#   bash tools/mesh_trace.sh design2 "7 9 10"
S=${1:-design2}; L=${2:-7 9 10}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/mtrace_$S -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_export.py --scene $S --levels $L --repeat 3 > /dev/null 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 - $S <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/mtrace_%s/**/t_kernel_trace.csv" % sys.argv[1], recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last export: from the last bbox kernel on
last = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("dcsg_bbox"))
out = {}
for r in rows[last:]:
    n = r["Kernel_Name"].split("(")[0]
    out.setdefault(n, []).append(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3))
print(sys.argv[1])
for n, v in out.items():
    print("  %-26s %7.3f ms  %s" % (n, sum(v), v if len(v) <= 10 else len(v)))
PY
