#!/bin/bash
# scripts/profile_round.sh <tag> -- on the GPU box: default bench, kernel-trace stats and the two PMC passes (separate
# runs, as the guide prescribes), all written under gpurun_out/<tag>/ for copying into profiles/.
set -e
tag=${1:-prof}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --output-format csv --kernel-trace --stats -d $out/kt -o kt -- python3 bench.py --no-cpu-baseline > $out/bench_kt.json 2> $out/kt.err
rocprofv3 --output-format csv --kernel-trace --stats -d $out/kt1 -o kt1 -- python3 bench.py --no-cpu-baseline --no-overlap > $out/bench_kt_one_lane.json 2> $out/kt1.err
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o f -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2> $out/fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $out/write -o w -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2> $out/write.err
find $out -name '*.csv' | head -40
