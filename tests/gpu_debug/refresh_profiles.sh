# Round-end measurement pass on the GPU box (run from the repo root).  Order matters: the PMC passes (counters only,
# separate rocprofv3 runs) come first and profiles/tree_traffic.json is rewritten from them ON THE BOX, stamped with the
# library's build id, so that the bench lines that follow carry counter traffic of the very build they time.
#   bash tests/gpu_debug/refresh_profiles.sh r03
set -e
R=${1:-r03}
mkdir -p gpurun_out/$R
export TMPDIR=/tmp
bash tests/gpu_debug/pmc_passes.sh ${R}_cfg2 > gpurun_out/$R/pmc_cfg2.log 2>&1
echo pmc cfg2 done
bash tests/gpu_debug/pmc_passes.sh ${R}_cfg5 --workload cfg5 > gpurun_out/$R/pmc_cfg5.log 2>&1
echo pmc cfg5 done
TRAFFIC_ONLY=1 bash tests/gpu_debug/pmc_passes.sh ${R}_4B --disc 2SPLIT4B > gpurun_out/$R/pmc_4B.log 2>&1
TRAFFIC_ONLY=1 bash tests/gpu_debug/pmc_passes.sh ${R}_cfg3 --workload cfg3 > gpurun_out/$R/pmc_cfg3.log 2>&1
echo pmc others done
python profiles/traffic_from_pmc.py gpurun_out/pmc_${R}_cfg2 gpurun_out/pmc_${R}_4B gpurun_out/pmc_${R}_cfg3 gpurun_out/pmc_${R}_cfg5 > gpurun_out/$R/traffic.log 2>&1
cp profiles/tree_traffic.json gpurun_out/$R/tree_traffic.json
python profiles/pmc_table.py gpurun_out/pmc_${R}_cfg2 > gpurun_out/$R/pmc_table.txt
python profiles/pmc_table.py gpurun_out/pmc_${R}_cfg5 > gpurun_out/$R/pmc_table_cfg5_kdvv.txt
echo tables done
python bench.py > gpurun_out/$R/bench.json 2> gpurun_out/$R/bench.err
echo headline done
python bench.py --disc 2SPLIT4B > gpurun_out/$R/bench_2SPLIT4B.json 2>> gpurun_out/$R/bench.err
python bench.py --workload cfg3 > gpurun_out/$R/bench_cfg3.json 2>> gpurun_out/$R/bench.err
python bench.py --workload cfg4 > gpurun_out/$R/bench_cfg4.json 2>> gpurun_out/$R/bench.err
python bench.py --workload cfg5 > gpurun_out/$R/bench_cfg5.json 2>> gpurun_out/$R/bench.err
python bench.py --workload inverse > gpurun_out/$R/bench_inverse.json 2>> gpurun_out/$R/bench.err
echo workloads done
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$R/prof -o $R --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined > $GRAFT_REPO_ROOT/gpurun_out/$R/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/$R/rocprof.err
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$R/prof5 -o ${R}_cfg5 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined > $GRAFT_REPO_ROOT/gpurun_out/$R/bench_cfg5_under_rocprof.json 2>> $GRAFT_REPO_ROOT/gpurun_out/$R/rocprof.err
cd $GRAFT_REPO_ROOT
echo rocprof done
