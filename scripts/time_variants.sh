#!/bin/bash
# runs the default bench (short) once per variant library in variants/ and prints the per-kernel times
for f in variants/liblmx_$1_*.so; do
  LMX_SO_PATH=$PWD/$f timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-extra --no-overlap --steps 20 --warmup 2 $EXTRA_BENCH_ARGS 2>gpurun_out/variant_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$f', ' '.join('%s=%.4f'%(a.replace('k_',''),b) for a,b in k.items() if b>0))" || echo "$f failed"
done
