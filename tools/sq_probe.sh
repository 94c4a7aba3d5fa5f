#!/bin/bash
# SQ counters of the bfloat16 top-1 bench line for a development build of the library:
#   bash tools/sq_probe.sh <tag> [path/to/lib.so]      -> gpurun_out/sqp_<tag>_{A,B}/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
tag=$1
[ -n "$2" ] && export ISPA_HIP_LIBRARY=$R/$2
cd /tmp && export TMPDIR=/tmp
A="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
B="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_BUSY_CYCLES"
for p in A B; do
  eval set_=\$$p
  timeout -s KILL 120 rocprofv3 --pmc $set_ --kernel-trace --output-format csv -d $OUT/sqp_${tag}_$p -- \
      python3 $R/bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/sqp_${tag}_$p.err || { tail -3 $OUT/sqp_${tag}_$p.err; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for p in "AB":
    for f in glob.glob("$OUT/sqp_${tag}_%s/**/*counter_collection.csv" % p, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "nn_filter" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print("$tag", k, sum(v) / len(v))
PY
