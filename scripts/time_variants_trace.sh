#!/bin/bash
# per-variant rocprofv3 kernel-trace stats (average duration per kernel instantiation matching $2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in variants/liblmx_$1_*.so; do
  export LMX_SO_PATH=$PWD/$f
  rm -rf gpurun_out/vt
  rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/vt -o vt -- python3 bench.py --no-cpu-baseline --no-overlap --no-events --steps 5 --warmup 2 > /dev/null 2>&1
  python3 - "$f" "$2" <<'PY'
import csv, sys, re
out = []
for r in csv.DictReader(open("gpurun_out/vt/vt_kernel_stats.csv")):
    if re.search(sys.argv[2], r["Name"]):
        m = re.search(r"(k_[a-z0-9_]+(<\d+>)?)", r["Name"])
        out.append("%s %.1f us" % (m.group(1), float(r["AverageNs"]) / 1e3))
print(sys.argv[1], " | ".join(sorted(out)))
PY
done
