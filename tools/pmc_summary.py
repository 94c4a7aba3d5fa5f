"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic.

Usage (after the passes of profiles/README.md were run into gpurun_out/pmc_<workload>_<COUNTER>/):
    python tools/pmc_summary.py gpurun_out profiles/r01_pmc_traffic.json
Traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: the counters are in KiB, and on
gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads.
"""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
out = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*_FETCH_SIZE"))):
    wl = os.path.basename(d)[len("pmc_"):-len("_FETCH_SIZE")]
    vals = collections.defaultdict(dict)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted(glob.glob(os.path.join(src, f"pmc_{wl}_{counter}", "*", "*counter_collection.csv")),
                       key=os.path.getmtime)
        if not files:
            continue
        agg, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(files[-1])):       # the newest run if the directory holds several
            name = r["Kernel_Name"]
            if "(anonymous namespace)::" not in name or "at::native" in name:
                continue
            short = name.split("(anonymous namespace)::")[1].split("(")[0].split("<")[0]
            agg[short] += float(r["Counter_Value"])
            cnt[short] += 1
        for k in agg:
            vals[k][counter] = agg[k] / cnt[k]
    out[wl] = {k: {"fetch_kib": round(v.get("FETCH_SIZE", 0.0), 1), "write_kib": round(v.get("WRITE_SIZE", 0.0), 1),
                   "traffic_bytes": round((2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024)}
               for k, v in vals.items()}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
