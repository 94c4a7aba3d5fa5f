"""Copies the summaries of tools/profile_all.sh from gpurun_out/ into profiles/.

    python tools/profile_collect.py gpurun_out profiles r01

Per workload tag: <prefix>_stats_<tag>_kernel_stats.csv (rocprofv3 --stats), <prefix>_stats_<tag>_bench.json (the bench
line of that same profiled run), <prefix>_pmc_<tag>_<COUNTERS>.csv (mean counter value per kernel),
<prefix>_pmc_traffic.json (tools/pmc_summary.py) and <prefix>_bench_default.json.
"""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

src, dst, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
if not glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    sys.exit(f"no rocprofv3 counter output under {src}/pmc_*: nothing collected, {dst} left as it is")
for d in sorted(glob.glob(os.path.join(src, "stats_*"))):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)[len("stats_"):]
    hits = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if hits:                                             # the newest run if the directory holds several
        shutil.copy(hits[-1], os.path.join(dst, f"{rnd}_stats_{tag}_kernel_stats.csv"))
    js = os.path.join(src, f"stats_{tag}.json")
    if os.path.exists(js):
        lines = [l for l in open(js).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(dst, f"{rnd}_stats_{tag}_bench.json"), "w").write(lines[-1] + "\n")
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[len("pmc_"):]
    hits = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not hits:
        continue
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(hits[-1])):
        if "(anonymous namespace)::" not in r["Kernel_Name"] or "at::native" in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"][:160], r["Counter_Name"])
        agg[key] += float(r["Counter_Value"])
        cnt[key] += 1
    with open(os.path.join(dst, f"{rnd}_pmc_{name}.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Mean_Value"])
        for key in sorted(agg):
            w.writerow([key[0], key[1], cnt[key], round(agg[key] / cnt[key], 1)])
# matrix-pipe utilisation per kernel from the SQ passes: SQ_VALU_MFMA_BUSY_CYCLES summed over the chip's 1024 SIMDs
# against the cycles the kernel lasted (GRBM_GUI_ACTIVE is the sum over the 8 XCDs)
util = {}
for f in sorted(glob.glob(os.path.join(dst, f"{rnd}_pmc_*_SQ.csv"))):
    tag = os.path.basename(f)[len(f"{rnd}_pmc_"):-len("_SQ.csv")]
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]][r["Counter_Name"]] = float(r["Mean_Value"])
    for kern, c in per.items():
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and c.get("GRBM_GUI_ACTIVE", 0) > 0:
            short = kern.split("::")[-1].split("(")[0]
            util.setdefault(tag, {})[short] = {
                "mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"], "kernel_cycles": c["GRBM_GUI_ACTIVE"] / 8,
                "mfma_util": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 4)}
if util:
    import json
    json.dump(util, open(os.path.join(dst, f"{rnd}_pmc_mfma_util.json"), "w"), indent=1, sort_keys=True)
subprocess.check_call([sys.executable, os.path.join(os.path.dirname(__file__), "pmc_summary.py"), src,
                       os.path.join(dst, f"{rnd}_pmc_traffic.json")], stdout=subprocess.DEVNULL)
bd = os.path.join(src, "bench_default.json")
if os.path.exists(bd):
    lines = [l for l in open(bd).read().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join(dst, f"{rnd}_bench_default.json"), "w").write(lines[-1] + "\n")
print("collected into", dst)
