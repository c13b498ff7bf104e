#!/bin/bash
# A/B of the level-0 spread image layouts k_refine gathers from: banded (default) vs flat (LMX_LS_FLAT=1); easy and busy scenes,
# three lanes (the default) and one lane.
for rep in 1 2; do
for flat in 0 1; do
  for tex in 0.6 1.0; do
    for lanes in "" "--no-overlap"; do
    if [ $flat = 1 ]; then export LMX_LS_FLAT=1; else unset LMX_LS_FLAT; fi
    timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-extra --steps 300 --warmup 10 --texture $tex $lanes 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('flat=$flat texture=$tex %-12s' % '$lanes', '%.0f frames/s %.4f ms/step' % (d['value'], d['ms_per_step']), ' '.join('%s=%.4f'%(a.replace('k_',''),b) for a,b in k.items() if b>0))" || echo failed
    done
  done
done
done
