#!/bin/bash
# rocprofv3 evidence for the kernels an EXPORT runs (run through gpurun from the repo root):
#   bash profiles/export_profile.sh design2 "7 9 10" r04_export_design2_1024
# Pass 1: kernel trace of five exports (durations: the first export of a process is the cold one).  Passes 2-4: counters,
# each group in its own run with --kernel-trace only.  Reduce with `python profiles/summarise_export.py <tag>`.
set -o pipefail
export TMPDIR=/tmp
S=$1; L=$2; TAG=$3
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 tools/bench_export.py --scene $S --levels $L --repeat 5 > $OUT/export.json 2> $OUT/trace.err || exit 1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- python3 tools/bench_export.py --scene $S --levels $L --repeat 3 > /dev/null 2> $OUT/pmc_sq.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o wr -- python3 tools/bench_export.py --scene $S --levels $L --repeat 3 > /dev/null 2> $OUT/pmc_write.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o rd -- python3 tools/bench_export.py --scene $S --levels $L --repeat 3 > /dev/null 2> $OUT/pmc_fetch.err || exit 1
echo "$TAG done"
