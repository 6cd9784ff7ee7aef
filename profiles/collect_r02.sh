# copies what tests/gpu_debug/refresh_profiles.sh r02 left under gpurun_out/ into the tracked profiles/r02_* files
# and regenerates the derived tables (run from the repo root)
set -e
R=gpurun_out/r02
cp $R/bench.json profiles/r02_bench.json
cp $R/bench_2SPLIT4B.json profiles/r02_bench_2SPLIT4B.json
cp $R/bench_cfg3.json profiles/r02_bench_cfg3_batch64.json
cp $R/bench_cfg4.json profiles/r02_bench_cfg4_bound_states.json
cp $R/bench_cfg5.json profiles/r02_bench_cfg5_kdvv.json
cp $R/bench_under_rocprof.json profiles/r02_bench_under_rocprof.json
cp $R/prof/r02_kernel_stats.csv profiles/r02_kernel_stats.csv
[ -f $R/inverse_timing.json ] && cp $R/inverse_timing.json profiles/r02_inverse_timing.json
[ -f $R/bench_inverse.json ] && cp $R/bench_inverse.json profiles/r02_bench_inverse.json
python profiles/pmc_table.py gpurun_out/pmc_r02_cfg2 > profiles/r02_pmc_table.txt
python profiles/traffic_from_pmc.py gpurun_out/pmc_r02_cfg2 gpurun_out/pmc_r02_4B gpurun_out/pmc_r02_cfg3 gpurun_out/pmc_r02_cfg5
