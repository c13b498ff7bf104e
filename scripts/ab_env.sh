#!/bin/bash
# Same-box A/B of one environment switch on the default bench workload: ROUNDS=3 scripts/ab_env.sh VAR v1 v2 [...]
var=$1; shift
rounds=${ROUNDS:-3}
for r in $(seq $rounds); do
  for v in "$@"; do
    env $var=$v python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %7.0f frames/s  %.4f ms/step ' % ('$var=$v', d['value'], d['ms_per_step']), {k: round(v, 4) for k, v in d['kernel_ms_per_step'].items() if v})"
  done
done
