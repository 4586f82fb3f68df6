#!/bin/bash
# Kernel trace of exports of one scene on the GPU box: per-launch durations of every kernel, in order.
#   bash tools/trace_export.sh flat256 "7 9 10" tag [repeat]
set -o pipefail
export TMPDIR=/tmp
S=$1; L=$2; TAG=$3; REP=${4:-2}
OUT=gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 tools/bench_export.py --scene $S --levels $L --repeat $REP > $OUT/export.json 2> $OUT/err.txt || exit 1
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[1] + "/launches.txt", "w") as o:
    for r in rows:
        o.write("%10.3f ms  %9.3f us  grid %8s  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                    r.get("Grid_Size", r.get("Grid_Size_X", "?")), r["Kernel_Name"][:60]))
PY
