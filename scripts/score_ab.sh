#!/bin/bash
# A/B of the scoring kernels on the bench workload: one lane (exclusive kernel times) and the default lanes
out=gpurun_out/${1:-score_ab}
mkdir -p $out
for v in u8 sb; do
  LMX_SCORE_KERNEL=$v python3 bench.py --no-extra --no-cpu-baseline --no-overlap --steps 200 > $out/one_lane_$v.json 2> $out/one_lane_$v.err
  LMX_SCORE_KERNEL=$v python3 bench.py --no-extra --no-cpu-baseline --steps 400 > $out/lanes_$v.json 2> $out/lanes_$v.err
  LMX_SCORE_KERNEL=$v python3 bench.py --no-extra --no-cpu-baseline --steps 200 --texture 1.0 > $out/busy_$v.json 2> $out/busy_$v.err
done
python3 - <<PY
import json
for v in ("u8", "sb"):
    for tag in ("one_lane", "lanes", "busy"):
        j = json.loads(open("$out/%s_%s.json" % (tag, v)).read().strip().splitlines()[-1])
        print("%-3s %-9s %9.0f frames/s  %.3f ms/step  score %.3f ms  kernels %s" % (v, tag, j["value"], j["ms_per_step"], j["kernel_ms_per_step"]["k_score_coarse"],
              {k: round(x, 3) for k, x in j["kernel_ms_per_step"].items() if x}))
PY
