#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   bash profiles/run_profile.sh r01 [bench flags...]
# Kernel-trace stats and each PMC group are separate runs (counters in their own pass).
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-export --no-secondary "$@" > $OUT/stats_bench.json 2> $OUT/stats.err || exit 1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-export --no-secondary "$@" > /dev/null 2> $OUT/pmc_sq.err || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-export --no-secondary "$@" > /dev/null 2> $OUT/pmc_sq2.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o wr -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-export --no-secondary "$@" > /dev/null 2> $OUT/pmc_write.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o rd -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-export --no-secondary "$@" > /dev/null 2> $OUT/pmc_fetch.err || exit 1
find $OUT -name "*.csv" | head -40
