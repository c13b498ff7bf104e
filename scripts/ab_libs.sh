#!/bin/bash
# Same-box A/B of two builds of liblmx.so (boxes differ by a few per cent, more than most kernel-level changes):
#   scripts/ab_libs.sh <liblmx.so A> <liblmx.so B> [rounds]     on the GPU box, e.g. through gpurun with build/<variant>/... in the tree
# alternates A and B: the default bench workload (--no-extra) and the many-candidates workloads (scripts/refine_load_profile.py quick).
a=$1; b=$2; rounds=${3:-2}
for r in $(seq $rounds); do
  for w in A B; do
    lib=$a; [ $w = B ] && lib=$b
    LMX_SO_PATH=$lib python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w default  %7.0f frames/s  %.4f ms/step ' % (d['value'], d['ms_per_step']), {k: round(v, 4) for k, v in d['kernel_ms_per_step'].items() if v})"
    LMX_SO_PATH=$lib python scripts/refine_load_profile.py - quick 2>/dev/null | grep -v "^    k_refine" | sed "s/^/$w /"
  done
done
