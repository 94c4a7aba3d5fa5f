"""Per-kernel SQ counter table from rocprofv3 --pmc passes (tools/sq_kernel.sh writes them).

    python tools/sq_summary.py <kernel substring> gpurun_out/sqk_<tag>_A gpurun_out/sqk_<tag>_B ... [--json out.json]

Dispatches of one kernel are grouped by grid size (a bench run launches the same kernel at several sizes);
the largest grid is reported.  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*, SQ_BUSY_CYCLES count quad-cycles,
SQ_VALU_MFMA_BUSY_CYCLES cycles, SQ_INSTS_* instructions (all summed over the chip)."""
import collections
import csv
import glob
import json
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--json")]
kern, dirs = args[0], args[1:]
out_json = None
if "--json" in sys.argv:
    out_json = sys.argv[sys.argv.index("--json") + 1]
    dirs = [d for d in dirs if d != out_json]
table = {}
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                per[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
                per[int(r["Grid_Size"])]["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        if not per:
            continue
        grid = max(per)
        for k, v in per[grid].items():
            table[k] = sum(v) / len(v)
        table["_grid"] = grid
for k in sorted(table):
    print(f"{k:32s} {table[k]:18.1f}")
if out_json:
    json.dump(table, open(out_json, "w"), indent=1, sort_keys=True)
