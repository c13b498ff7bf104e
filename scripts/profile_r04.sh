#!/bin/bash
# scripts/profile_r04.sh <tag> -- on the GPU box: scripts/profile_r02.sh <tag> (the default bench line, rocprofv3 kernel-trace stats of the main workload
# with three lanes / one lane / the whole default command, one --pmc pass per counter group on the one-lane bench), then the same one-lane
# kernel trace and counter passes with LMX_SCORE_NO_PRUNE=1 -- the scoring kernel's data-independent full-work leg (bench.py extra.score_full_work)
# -- under gpurun_out/<tag>/fw/.  scripts/pmc_summary2.py turns both into the tables committed under profiles/.
tag=${1:-r04}
root=$PWD
bash scripts/profile_r02.sh $tag || exit 1
out=$root/gpurun_out/$tag/fw
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
export LMX_SCORE_NO_PRUNE=1
python3 bench.py --no-cpu-baseline --no-extra --no-overlap > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
rocprofv3 --output-format csv --kernel-trace --stats -d $out/kt1 -o kt1 -- python3 bench.py --no-cpu-baseline --no-extra --no-overlap > $out/bench_kt_one_lane.json 2> $out/kt1.err || exit 1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $set --kernel-trace -d $out/pmc$i -o p -- python3 bench.py --no-cpu-baseline --no-overlap --no-events --no-extra --steps 4 --warmup 1 > /dev/null 2> $out/pmc$i.err || echo "PMC pass $i ($set) failed"
done
python3 scripts/pmc_summary2.py $out > $out/summary.txt && grep -A12 "k_score_coarse" $out/summary.txt | head -20
