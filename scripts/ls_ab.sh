#!/bin/bash
# A/B of the level-0 spread image layout (flat vs column-blocked) on the bench workload, easy and busy scenes, one lane and three
out=gpurun_out/${1:-ls_ab}
mkdir -p $out
for v in flat tiled; do
  if [ $v = flat ]; then export LMX_LS_FLAT=1; else unset LMX_LS_FLAT; fi
  for tag in "easy --texture 0.6" "busy --texture 1.0"; do
    set -- $tag
    python3 bench.py --no-extra --no-cpu-baseline --no-overlap --steps 200 $2 $3 > $out/${v}_$1_one.json 2> $out/${v}_$1_one.err
    python3 bench.py --no-extra --no-cpu-baseline --steps 300 $2 $3 > $out/${v}_$1_lanes.json 2> $out/${v}_$1_lanes.err
  done
done
python3 - <<PY
import json
for v in ("flat", "tiled"):
    for s in ("easy", "busy"):
        for l in ("one", "lanes"):
            j = json.loads(open("$out/%s_%s_%s.json" % (v, s, l)).read().strip().splitlines()[-1])
            print("%-5s %-4s %-5s %9.0f frames/s  %.4f ms/step  %s" % (v, s, l, j["value"], j["ms_per_step"], {k[2:]: round(x, 3) for k, x in j["kernel_ms_per_step"].items() if x}))
PY
