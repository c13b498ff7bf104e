#!/bin/bash
# Same-box A/B of two builds of liblmx.so on the default bench workload only (boxes differ by a few per cent: compare on ONE box, alternating):
#   scripts/ab_quick.sh <liblmx.so A> <liblmx.so B> [rounds]      prints frames/s, ms/step and the per-kernel ms of every run
a=$1; b=$2; rounds=${3:-3}
for r in $(seq $rounds); do
  for w in A B; do
    lib=$a; [ $w = B ] && lib=$b
    LMX_SO_PATH=$lib python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w %s  %7.0f frames/s  %.4f ms/step ' % ('$lib'.split('/')[-1], d['value'], d['ms_per_step']), {k: round(v, 4) for k, v in d['kernel_ms_per_step'].items() if v})"
  done
done
