#!/bin/bash
# occupancy caps of the issue-bound kernels vs throughput with three lanes (see lds_pad in lmx_kernels.hip)
out=gpurun_out/${1:-pad}
mkdir -p $out
run() {  # name, env...
  name=$1; shift
  env "$@" python3 bench.py --no-extra --no-cpu-baseline --steps 300 > $out/$name.json 2> $out/$name.err
  python3 - <<PY
import json
j = json.loads(open("$out/$name.json").read().strip().splitlines()[-1])
print("%-28s %9.0f frames/s  %.4f ms/step  kernels %s" % ("$name", j["value"], j["ms_per_step"], {k[2:]: round(x, 3) for k, x in j["kernel_ms_per_step"].items() if x}))
PY
}
run base
run color13k LMX_LDS_PAD_COLOR=13000
run color7k LMX_LDS_PAD_COLOR=7000
run depth13k LMX_LDS_PAD_DEPTH=13400
run cd13k LMX_LDS_PAD_COLOR=13000 LMX_LDS_PAD_DEPTH=13400
run cd7k LMX_LDS_PAD_COLOR=7000 LMX_LDS_PAD_DEPTH=7400
run cds13k LMX_LDS_PAD_COLOR=13000 LMX_LDS_PAD_DEPTH=13400 LMX_LDS_PAD_SPREAD=19000
run cd20k LMX_LDS_PAD_COLOR=20500 LMX_LDS_PAD_DEPTH=21000
