# copies what tests/gpu_debug/refresh_profiles.sh r03 left under gpurun_out/ into the tracked profiles/r03_* files
# and regenerates the derived tables (run from the repo root)
set -e
R=gpurun_out/r03
cp $R/tree_traffic.json profiles/tree_traffic.json
cp $R/bench.json profiles/r03_bench.json
cp $R/bench_2SPLIT4B.json profiles/r03_bench_2SPLIT4B.json
cp $R/bench_cfg3.json profiles/r03_bench_cfg3_batch64.json
cp $R/bench_cfg4.json profiles/r03_bench_cfg4_bound_states.json
cp $R/bench_cfg5.json profiles/r03_bench_cfg5_kdvv.json
cp $R/bench_inverse.json profiles/r03_bench_inverse.json
cp $R/bench_under_rocprof.json profiles/r03_bench_under_rocprof.json
cp $R/bench_cfg5_under_rocprof.json profiles/r03_bench_cfg5_under_rocprof.json
cp $R/prof/r03_kernel_stats.csv profiles/r03_kernel_stats.csv
cp $R/prof5/r03_cfg5_kernel_stats.csv profiles/r03_kernel_stats_cfg5_kdvv.csv
cp $R/pmc_table.txt profiles/r03_pmc_table.txt
cp $R/pmc_table_cfg5_kdvv.txt profiles/r03_pmc_table_cfg5_kdvv.txt
python profiles/summarize.py profiles/r03_kernel_stats.csv profiles/r03_bench_under_rocprof.json > profiles/r03_rocprof_summary.md
python profiles/summarize.py profiles/r03_kernel_stats_cfg5_kdvv.csv profiles/r03_bench_cfg5_under_rocprof.json > profiles/r03_rocprof_summary_cfg5_kdvv.md
