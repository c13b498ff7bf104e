#!/bin/bash
# Same-box A/B/... of several builds of liblmx.so on the default bench workload only (boxes differ by a few per cent: compare on ONE box, alternating):
#   ROUNDS=3 scripts/ab_quick.sh <liblmx.so> <liblmx.so> [...]      prints frames/s, ms/step and the per-kernel ms of every run
rounds=${ROUNDS:-3}
for r in $(seq $rounds); do
  for lib in "$@"; do
    LMX_SO_PATH=$lib python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-18s %7.0f frames/s  %.4f ms/step ' % ('$lib'.split('/')[-1], d['value'], d['ms_per_step']), {k: round(v, 4) for k, v in d['kernel_ms_per_step'].items() if v})"
  done
done
