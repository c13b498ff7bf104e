#!/bin/bash
# scripts/profile_r02.sh <tag> [extra bench args] -- on the GPU box: the default bench line, rocprofv3 kernel-trace stats of the
# same command (default lanes and one lane), and the PMC passes (each its own run with --kernel-trace only, as the guide and
# gpurun require) on the one-lane bench.  Everything lands under gpurun_out/<tag>/; scripts/pmc_summary2.py turns it into the
# per-(kernel, grid) tables that are committed under profiles/.
tag=${1:-r02}
shift
extra="$@"
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# the queue count bench.py would set for itself, here in the environment of every command: under rocprofv3 the runtime is already up when
# bench.py starts (the r03 profiles were taken with the runtime default of 4; advisor finding)
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
cd $root
set -x
python3 bench.py $extra > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
# kernel-trace stats of the MAIN workload (--no-extra: the secondary lines launch the same kernels on other workloads -- busy scene,
# threshold 50, host frames -- and would be averaged into the same rows), default lanes and one lane; then of the full default command
rocprofv3 --output-format csv --kernel-trace --stats -d $out/kt -o kt -- python3 bench.py --no-cpu-baseline --no-extra $extra > $out/bench_kt.json 2> $out/kt.err || exit 1
rocprofv3 --output-format csv --kernel-trace --stats -d $out/kt1 -o kt1 -- python3 bench.py --no-cpu-baseline --no-extra --no-overlap $extra > $out/bench_kt_one_lane.json 2> $out/kt1.err || exit 1
rocprofv3 --output-format csv --kernel-trace --stats -d $out/ktfull -o ktfull -- python3 bench.py --no-cpu-baseline $extra > $out/bench_ktfull.json 2> $out/ktfull.err || exit 1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_IOPS SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_IFETCH SQ_INSTS_VALU_CVT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $set --kernel-trace -d $out/pmc$i -o p -- python3 bench.py --no-cpu-baseline --no-overlap --no-events --no-extra --steps 4 --warmup 1 $extra > /dev/null 2> $out/pmc$i.err || echo "PMC pass $i ($set) failed"
done
set +x
python3 scripts/pmc_summary2.py $out > $out/summary.txt && cat $out/summary.txt
