#!/bin/bash
# scripts/pmc_sq.sh <tag>: SQ instruction-mix / stall counters of every kernel (separate rocprofv3 --pmc passes, short bench)
tag=${1:-sq}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $set --kernel-trace -d $out/p$i -o p -- python3 bench.py --no-cpu-baseline --no-overlap --no-events --steps 3 --warmup 1 > /dev/null 2> $out/p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections, re, statistics
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if not m: continue
        acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
for k in sorted(acc):
    print(k)
    for c in names:
        if acc[k].get(c): print("   %-26s %14.4g" % (c, statistics.median(acc[k][c])))
PY
