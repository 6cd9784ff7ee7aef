# PMC passes of the headline bench (separate rocprofv3 runs, counters only -- never together with tracing).
# TRAFFIC_ONLY=1 skips the SQ passes (FETCH_SIZE / WRITE_SIZE only).
# usage: bash tests/gpu_debug/pmc_passes.sh TAG [bench args]; output under gpurun_out/pmc_TAG/
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"
if [ -z "$TRAFFIC_ONLY" ]; then
rocprofv3 --pmc $P1 -d $OUT/p1 -o p1 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined --no-host-call "$@" > $OUT/p1.json 2> $OUT/p1.err
echo p1 done
rocprofv3 --pmc $P2 -d $OUT/p2 -o p2 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined --no-host-call "$@" > $OUT/p2.json 2> $OUT/p2.err
echo p2 done
fi
rocprofv3 --pmc FETCH_SIZE -d $OUT/p3 -o p3 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined --no-host-call "$@" > $OUT/p3.json 2> $OUT/p3.err
echo p3 done
rocprofv3 --pmc WRITE_SIZE -d $OUT/p4 -o p4 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined --no-host-call "$@" > $OUT/p4.json 2> $OUT/p4.err
echo p4 done
