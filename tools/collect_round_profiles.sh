# Round evidence on the GPU box: rocprofv3 stats + PMC passes of bench.py for four workloads, the default bench line,
# and the export wall-clock table.  bash tools/collect_round_profiles.sh  (then profiles/summarise.py <tag> here)
set -o pipefail
bash profiles/run_profile.sh r01 > gpurun_out/p1.log 2>&1 && \
bash profiles/run_profile.sh r01_design2 --workload design2_512 > gpurun_out/p2.log 2>&1 && \
bash profiles/run_profile.sh r01_flat256 --workload flat256_512 > gpurun_out/p3.log 2>&1 && \
bash profiles/run_profile.sh r01_grouped256 --workload grouped256_512 > gpurun_out/p4.log 2>&1 && \
python bench.py > gpurun_out/bench_r01_default.json 2> gpurun_out/bench_err.log && \
python tools/bench_export.py --scene sphere --cpu > gpurun_out/export_r01_sphere_512.json 2> gpurun_out/export_err.log && \
python tools/bench_export.py --scene design2 --cpu > gpurun_out/export_r01_design2_512.json 2>> gpurun_out/export_err.log && \
python tools/bench_export.py --scene design2 --levels 7 9 10 --cpu > gpurun_out/export_r01_design2_1024.json 2>> gpurun_out/export_err.log && \
python tools/bench_export.py --scene flat256 --levels 6 8 9 > gpurun_out/export_r01_flat256_512.json 2>> gpurun_out/export_err.log && \
python tools/bench_export.py --scene flat256 --levels 7 9 10 > gpurun_out/export_r01_flat256_1024.json 2>> gpurun_out/export_err.log && \
python tools/bench_export.py --scene grouped256 --levels 6 8 9 > gpurun_out/export_r01_grouped256_512.json 2>> gpurun_out/export_err.log && \
python tools/bench_export.py --scene grouped256 --levels 7 9 10 > gpurun_out/export_r01_grouped256_1024.json 2>> gpurun_out/export_err.log
echo done $?
