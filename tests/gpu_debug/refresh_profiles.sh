# Round-end measurement pass on the GPU box (run from the repo root): bench lines of every workload, rocprofv3 kernel
# statistics of the headline bench, PMC passes (counters only) of the headline and of the other workloads' trees.
#   bash tests/gpu_debug/refresh_profiles.sh r02
set -e
R=${1:-r02}
mkdir -p gpurun_out/$R
export TMPDIR=/tmp
python bench.py > gpurun_out/$R/bench.json 2> gpurun_out/$R/bench.err
echo headline done
python bench.py --disc 2SPLIT4B > gpurun_out/$R/bench_2SPLIT4B.json 2>> gpurun_out/$R/bench.err
python bench.py --workload cfg3 > gpurun_out/$R/bench_cfg3.json 2>> gpurun_out/$R/bench.err
python bench.py --workload cfg4 > gpurun_out/$R/bench_cfg4.json 2>> gpurun_out/$R/bench.err
python bench.py --workload cfg5 > gpurun_out/$R/bench_cfg5.json 2>> gpurun_out/$R/bench.err
echo workloads done
rocprofv3 --kernel-trace --stats -d gpurun_out/$R/prof -o $R --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined > gpurun_out/$R/bench_under_rocprof.json 2> gpurun_out/$R/rocprof.err
echo rocprof done
bash tests/gpu_debug/pmc_passes.sh ${R}_cfg2 > gpurun_out/$R/pmc_cfg2.log 2>&1
echo pmc cfg2 done
TRAFFIC_ONLY=1 bash tests/gpu_debug/pmc_passes.sh ${R}_4B --disc 2SPLIT4B > gpurun_out/$R/pmc_4B.log 2>&1
TRAFFIC_ONLY=1 bash tests/gpu_debug/pmc_passes.sh ${R}_cfg3 --workload cfg3 > gpurun_out/$R/pmc_cfg3.log 2>&1
TRAFFIC_ONLY=1 bash tests/gpu_debug/pmc_passes.sh ${R}_cfg5 --workload cfg5 > gpurun_out/$R/pmc_cfg5.log 2>&1
echo pmc others done
