# Round evidence on the GPU box: rocprofv3 stats + PMC passes of bench.py for the bench workloads (+ the relaxed-numerics
# module and the LDS-table interpreter on the headline workload), the default bench line, the export wall-clock table
# (stage times, and scene files -> closed outputs from a fresh process, warm and cold JIT) and a kernel trace of two
# exports.  Two parts, each within one gpurun call (20 minutes):
#   bash tools/collect_round_profiles.sh a ; bash tools/collect_round_profiles.sh b ; bash tools/collect_round_profiles.sh c
# then `python profiles/summarise.py <tag>` here for every tag.
set -o pipefail
export TMPDIR=/tmp
R=r04
PART=${1:-a}
if [ "$PART" = a ]; then
bash profiles/run_profile.sh $R > gpurun_out/p1.log 2>&1 && \
bash profiles/run_profile.sh ${R}_sphere1024 --workload sphere1024 > gpurun_out/p1b.log 2>&1 && \
bash profiles/run_profile.sh ${R}_design2 --workload design2_512 > gpurun_out/p2.log 2>&1 && \
bash profiles/run_profile.sh ${R}_design2_1024 --workload design2_1024 --steps 4 > gpurun_out/p2b.log 2>&1 && \
DCSG_SPECIALISE=0 bash profiles/run_profile.sh ${R}_s4 > gpurun_out/p7.log 2>&1 && \
DCSG_NUMERICS=relaxed bash profiles/run_profile.sh ${R}_relaxed > gpurun_out/p6.log 2>&1 || exit 1
echo "part a done"
exit 0
fi
if [ "$PART" = b ]; then
bash profiles/run_profile.sh ${R}_flat256 --workload flat256_512 > gpurun_out/p3.log 2>&1 && \
bash profiles/run_profile.sh ${R}_grouped256 --workload grouped256_512 > gpurun_out/p4.log 2>&1 && \
bash profiles/run_profile.sh ${R}_smooth256 --workload smooth256_512 > gpurun_out/p5.log 2>&1 && \
bash profiles/run_profile.sh ${R}_userbrush128 --workload userbrush128_512 --steps 5 > gpurun_out/p5b.log 2>&1 && \
python bench.py > gpurun_out/bench_${R}_default.json 2> gpurun_out/bench_err.log || exit 1
: > gpurun_out/export_err.log
for cfg in "sphere 6 8 9 --cpu" "design2 6 8 9 --cpu" "design2 7 9 10 --cpu" "flat256 6 8 9" "flat256 7 9 10" \
           "grouped256 6 8 9" "grouped256 7 9 10" "smooth256 6 8 9" "smooth256 7 9 10"; do
  set -- $cfg
  size=$((1 << $4))
  python tools/bench_export.py --scene $1 --levels $2 $3 $4 $5 --repeat 5 > gpurun_out/export_${R}_$1_$size.json 2>> gpurun_out/export_err.log || exit 1
done
rm -f gpurun_out/export_${R}_from_disk.jsonl
for s in design2 flat256; do
  for l in "6 8 9" "7 9 10"; do
    for a in "" "--cold-jit"; do
      python tools/export_from_disk.py --scene-dir tests/golden/scenes/$s --levels $l $a >> gpurun_out/export_${R}_from_disk.jsonl 2>> gpurun_out/export_err.log || exit 1
    done
  done
done
echo "part b done"
exit 0
fi
if [ "$PART" = c ]; then
# kernel trace + counters of the exports themselves (profiles/export_profile.sh; reduce with profiles/summarise_export.py)
for cfg in "design2 6 8 9" "design2 7 9 10" "flat256 7 9 10" "grouped256 7 9 10" "smooth256 7 9 10" "userbrush128 6 8 9"; do
  set -- $cfg
  bash profiles/export_profile.sh $1 "$2 $3 $4" ${R}_export_$1_$((1 << $4)) > gpurun_out/pe_$1_$4.log 2>&1 || exit 1
done
echo "part c done"
exit 0
fi
