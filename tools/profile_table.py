"""Markdown table of the committed rocprofv3 evidence of one profiling pass (file prefix, e.g. r03):
    python tools/profile_table.py profiles r03
Per profiled bench command: the bench line of the traced run (value, step time, kernel time by HIP events), the
traced average of the dominant kernel, the algorithmic work per launch and the roofline fraction it gives with the
TRACED duration, and the HBM bytes of the counter passes.  Profiled runs are a few per cent slower than plain ones
(the default bench line of the same tree is <prefix>_bench_default.json)."""
import csv
import glob
import json
import os
import sys

dst, rnd = sys.argv[1], sys.argv[2]
traffic = json.load(open(os.path.join(dst, f"{rnd}_pmc_traffic.json")))
util = json.load(open(os.path.join(dst, f"{rnd}_pmc_mfma_util.json")))
KERNEL = {"nn_filter": ("nn_filter",), "project_prune": ("project_prune2_kernel",), "zbuffer_tiled": ("zb_tile_kernel",),
          "zbuffer": ("zbuffer_kernel",), "project": ("project2_kernel",),
          "make_disparity_ws": ("disparity_",),
          "corr_eccv": ("corr_eccv_tile_kernel",), "corr_acmm_gemm": ("corr_acmm_gemm_kernel",)}
print("| bench command (`--workload ...`) | value | step ms | dominant kernel(s), traced avg | work per launch | achieved (traced) | frac of peak | counter HBM bytes | matrix-pipe busy |")
print("|---|---|---|---|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(dst, f"{rnd}_stats_*_bench.json"))):
    tag = os.path.basename(f)[len(f"{rnd}_stats_"):-len("_bench.json")]
    rec = json.loads(open(f).read())
    roof = rec["roofline"]
    stats = {}
    for r in csv.DictReader(open(os.path.join(dst, f"{rnd}_stats_{tag}_kernel_stats.csv"))):
        if "(anonymous namespace)::" not in r["Name"] or "at::native" in r["Name"]:
            continue
        short = r["Name"].split("(anonymous namespace)::")[1].split("(")[0].split("<")[0]
        stats[short] = max(stats.get(short, 0.0), float(r["AverageNs"]) * 1e-6)      # ms; largest instantiation
    names = KERNEL.get(roof["kernel"], (roof["kernel"],))
    hits = [(k, v) for k, v in stats.items() if any(k.startswith(n) for n in names)]
    # a bench run may launch a filter kernel at other sizes too (warm-up of other paths): the longest is the one
    if roof["kernel"] == "nn_filter":
        hits = [max(hits, key=lambda kv: kv[1])]
    traced = sum(v for _, v in hits)
    work = roof.get("flop_per_launch", roof.get("work_per_launch"))
    unit = roof["unit"]
    ach = work / (traced * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9)
    tb = sum(traffic.get(tag, {}).get(k, {}).get("traffic_bytes", 0) for k, _ in hits)
    mu = [v["mfma_util"] for k, v in util.get(tag, {}).items()]
    print(f"| `{tag.replace('___', ' --').replace('_', ' ')}` | {rec['value']:.0f} {rec['unit']} | {rec['ms_per_step']:.3f} | "
          f"{' + '.join(k for k, _ in hits)} {traced:.3f} ms | {work:.4g} {'FLOP' if unit == 'TFLOP/s' else 'B'} | "
          f"{ach:.1f} {unit} | {ach / roof['peak']:.3f} | {tb / 1e9:.2f} GB | {mu[0] if mu else '-'} |")
