# Round evidence on the GPU box: rocprofv3 stats + PMC passes of bench.py for five workloads, the default bench line,
# the export wall-clock table and a kernel trace of two exports.
#   bash tools/collect_round_profiles.sh      (then `python profiles/summarise.py <tag>` here for every tag)
set -o pipefail
export TMPDIR=/tmp
bash profiles/run_profile.sh r02 > gpurun_out/p1.log 2>&1 && \
bash profiles/run_profile.sh r02_design2 --workload design2_512 > gpurun_out/p2.log 2>&1 && \
bash profiles/run_profile.sh r02_flat256 --workload flat256_512 > gpurun_out/p3.log 2>&1 && \
bash profiles/run_profile.sh r02_grouped256 --workload grouped256_512 > gpurun_out/p4.log 2>&1 && \
bash profiles/run_profile.sh r02_smooth256 --workload smooth256_512 > gpurun_out/p5.log 2>&1 && \
python bench.py > gpurun_out/bench_r02_default.json 2> gpurun_out/bench_err.log || exit 1
: > gpurun_out/export_err.log
for cfg in "sphere 6 8 9 --cpu" "design2 6 8 9 --cpu" "design2 7 9 10 --cpu" "flat256 6 8 9" "flat256 7 9 10" \
           "grouped256 6 8 9" "grouped256 7 9 10" "smooth256 6 8 9" "smooth256 7 9 10"; do
  set -- $cfg
  size=$((1 << $4))
  python tools/bench_export.py --scene $1 --levels $2 $3 $4 $5 > gpurun_out/export_r02_$1_$size.json 2>> gpurun_out/export_err.log || exit 1
done
for s in design2 flat256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_export_$s -o x -- python3 tools/bench_export.py --scene $s --levels 7 9 10 --repeat 3 > /dev/null 2>> gpurun_out/export_err.log || exit 1
done
echo done
