#!/bin/bash
# SQ counters of one kernel (name substring) in any bench command:
#   bash tools/sq_kernel.sh <tag> <kernel substring> <bench args...>     -> gpurun_out/sqk_<tag>_{A,B}/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
tag=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp
A="GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES"
B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS"
for p in A B; do
  eval set_=\$$p
  timeout -s KILL 120 rocprofv3 --pmc $set_ --kernel-trace --output-format csv -d $OUT/sqk_${tag}_$p -- \
      python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/sqk_${tag}_$p.err || { tail -3 $OUT/sqk_${tag}_$p.err; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for p in "AB":
    for f in glob.glob("$OUT/sqk_${tag}_%s/**/*counter_collection.csv" % p, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "$kern" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print("$tag", k, sum(v) / len(v))
PY
