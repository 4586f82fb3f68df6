#!/bin/bash
# One counter pass of bench.py (counters from $DCSG_PMC; diagnostic; run through gpurun from the repo root): bash tools/pmc_quick.sh TAG [bench flags]
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/pmcq_$TAG
mkdir -p $OUT
CTRS=${DCSG_PMC:-"SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR"}
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-export --no-secondary "$@" > /dev/null 2> $OUT/err.txt || exit 1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "sweep" not in k: continue
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
