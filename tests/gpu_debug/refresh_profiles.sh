# Round-end measurement pass on the GPU box: smoke, the bench lines of every workload, and the rocprofv3
# kernel statistics of the headline bench.  Run from the repo root: bash tests/gpu_debug/refresh_profiles.sh
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r01b_smoke.log 2>&1
python bench.py > gpurun_out/r01b_bench.json 2> gpurun_out/r01b_bench.err
echo bench1 done
python bench.py --disc 2SPLIT4B > gpurun_out/r01b_bench_2SPLIT4B.json 2>> gpurun_out/r01b_bench.err
echo bench4B done
python bench.py --workload cfg3 > gpurun_out/r01b_bench_cfg3.json 2>> gpurun_out/r01b_bench.err
echo cfg3 done
python bench.py --workload cfg5 > gpurun_out/r01b_bench_cfg5.json 2>> gpurun_out/r01b_bench.err
echo cfg5 done
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01b -o r01b --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r01b_bench_under_rocprof.json 2> gpurun_out/r01b_rocprof.err
echo rocprof done
cat gpurun_out/r01b_bench.json
