#!/bin/bash
# VALU/SALU/LDS instructions per wave of every kernel (one rocprofv3 --pmc pass, short one-lane bench)
out=$PWD/gpurun_out/valu
rm -rf $out; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace -d $out -o p -- python3 bench.py --no-cpu-baseline --no-overlap --no-events --steps 3 --warmup 1 > /dev/null 2> $out/err.txt
python3 - <<PY
import csv, collections, re, statistics
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$out/p_counter_collection.csv")):
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    if m: acc[(m.group(1), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    w = statistics.median(acc[k]["SQ_WAVES"])
    if w < 1000: continue
    g = lambda c: statistics.median(acc[k][c]) / w
    print("%-24s grid %9s waves %7d | per wave: VALU %5.0f SALU %5.0f LDS %4.0f lifetime %6.0f quads" % (k[0], k[1], w, g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_WAVE_CYCLES")))
PY
