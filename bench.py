#!/usr/bin/env python3
"""Throughput bench of the dense correspondence stage (BASELINE.json's metric).

A step = one pass of the hot path over one batch of synthetic image pairs resident in HBM:
dense nearest-neighbour correspondence between a 64x64 grid of 128-d query features and a
64x64 grid of template features per pair (ispa_nn_topk through the C ABI: norms, MFMA filter,
finalize, float64 exact pass).  With N GPUs each rank holds its own batch (weak scaling, no
collective in the data path) and the index/distance records are all-gathered once per step.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  The default run (one GPU, no flags) also carries, under `extra`, side
measurements that are not part of `value`: the bfloat16 and split-bfloat16 filters, queries with close matches, and
(`extra.stages`) the projection, z-buffer, disparity and correlation stages at their BASELINE sizes, each through
the same code as `--workload <stage>`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}       # dense peaks, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=512, help="image pairs per GPU per step")
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--workload", choices=["nn", "project", "disparity", "zbuffer", "corr"], default="nn",
                    help="nn: dense NN correspondence (the headline); project: 100k points x 3600 poses "
                         "projection + visibility + crop; disparity: make_disparity flow maps of 100k-point renders")
    ap.add_argument("--poses", type=int, default=3600)
    ap.add_argument("--maps", type=int, default=256)
    ap.add_argument("--points", type=int, default=100000)
    ap.add_argument("--one-launch", action="store_true",
                    help="disparity: the one-launch kernel without a workspace (ispa_make_disparity) instead of the "
                         "form with a workspace (ispa_make_disparity_ws)")
    ap.add_argument("--model", choices=["chair", "cube"], default="chair",
                    help="geometry workloads: the synthetic model the points come from. chair: area-weighted surface "
                         "samples of a box-built chair mesh (the reference samples CAD mesh surfaces, "
                         "run_proj.py:15-22); cube: points uniform in a solid cube (no surface: the worst case for "
                         "anything that relies on neighbouring points landing on neighbouring pixels)")
    ap.add_argument("--unfused", action="store_true",
                    help="disparity workload through project_prune x2 + flow_splat instead of the fused kernel")
    ap.add_argument("--filter", choices=["native", "split"], default="native",
                    help="f32 only: 'split' runs the filter pass on bfloat16 MFMAs over a hi/lo split of the "
                         "float32 inputs (same exact results)")
    ap.add_argument("--no-gather", action="store_true", help="skip the per-step result all-gather")
    ap.add_argument("--matches", action="store_true",
                    help="nn: queries are noisy copies of (permuted) template cells, as for two views of one object: "
                         "every nearest distance is small, so every returned distance takes the float64 refinement")
    ap.add_argument("--sorted", "--morton", dest="morton", action="store_true",
                    help="geometry workloads: put the model points in space-filling-curve order first "
                         "(run_proj.spatial_order, once per model), so that neighbouring lanes hit neighbouring pixels")
    ap.add_argument("--as-generated", action="store_true",
                    help="disparity: process the points in the order they were generated in (ispa_make_disparity_ws) "
                         "instead of along a space-filling curve with the generated order as ranks "
                         "(ispa_make_disparity_ordered; same maps)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend; 'gloo' with ranks sharing one GPU rehearses the multi-rank "
                         "code path on a single-GPU box (not a measurement)")
    ap.add_argument("--depth-per-pose", action="store_true",
                    help="project: every pose tests visibility against a depth map of its own (P x 540 x 960 float32, "
                         "as make_disparity uses them: one render per pose) instead of one map shared by all poses, "
                         "so that the depth gathers come from HBM rather than from L2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the side measurements of the default run (bfloat16 / split filter / close matches, and "
                         "the projection, z-buffer, disparity and correlation stages at their BASELINE sizes)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median_rate(fn, units_per_call, budget_s, repeats=5, max_calls=400):
    """`repeats` timed batches of calls within the budget -> (median units/s, calls per batch, seconds used)."""
    fn()                                                    # warm
    t0 = time.perf_counter()
    fn()
    one = max(time.perf_counter() - t0, 1e-6)
    per = int(max(1, min(max_calls // repeats, budget_s / repeats / one)))
    rates, used = [], 0.0
    for _ in range(repeats):
        t0 = time.perf_counter()
        for _ in range(per):
            fn()
        el = time.perf_counter() - t0
        used += el
        rates.append(units_per_call * per / el)
    return float(np.median(rates)), per, used


def cpu_baseline(grid, channels, k, budget_s):
    """numpy float32 formulation (sgemm + |a|^2+|b|^2-2ab + argmin) on the host cores: all BLAS threads
    (the headline figure) and pinned to one thread; five timed batches each, median."""
    from oracle import nn as onn
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        cores = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    except Exception:
        threadpool_limits = None
        cores = os.cpu_count() or 1
    rs = np.random.RandomState(2)
    n = grid * grid
    q = rs.randn(channels, n).astype(np.float32)
    r = rs.randn(channels, n).astype(np.float32)
    q /= np.sqrt((q ** 2).sum(0, keepdims=True))
    r /= np.sqrt((r ** 2).sum(0, keepdims=True))
    rate, per, used = _median_rate(lambda: onn.nn_sgemm(q, r, k), 1, 0.7 * budget_s)
    out = {"value": round(rate, 3), "unit": "pairs/s", "cores": int(cores), "kind": "port", "cpu_model": cpu_model(),
           "repeats": 5, "statistic": "median",
           "sample": f"5 batches of {per} pairs of {grid}x{grid}x{channels} float32 in {used:.1f} s, numpy sgemm + "
                     f"argmin (oracle.nn.nn_sgemm), BLAS threads = {cores}"}
    if threadpool_limits is not None:
        with threadpool_limits(limits=1, user_api="blas"):
            rate1, per1, used1 = _median_rate(lambda: onn.nn_sgemm(q, r, k), 1, 0.3 * budget_s)
        out["one_core"] = {"value": round(rate1, 3), "unit": "pairs/s", "cores": 1,
                           "sample": f"5 batches of {per1} pairs in {used1:.1f} s, BLAS pinned to 1 thread"}
    return out


HBM_PEAK_GBS = 8000.0                                     # MI355X_MICROARCH.md


def measured_traffic(workload_key, kernel, is_default_shape):
    """HBM bytes per launch of `kernel` and the file they come from: the newest committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in separate runs of this same bench command, gfx950 correction applied by
    tools/pmc_summary.py).  A constant from that profile, NOT a measurement of this run; (None, None) when this
    run's shape is not the profiled one."""
    if not is_default_shape:
        return None, None
    try:
        import glob
        newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]
        table = json.load(open(newest))[workload_key]
        src = os.path.relpath(newest, ROOT)
        if kernel.endswith("*"):                         # the timed launches of one call: their traffic adds up
            hits = [v["traffic_bytes"] for k, v in table.items() if k.startswith(kernel[:-1])]
            return (sum(hits), src) if hits else (None, None)
        if kernel in table:
            return table[kernel]["traffic_bytes"], src
        hits = [v["traffic_bytes"] for k, v in table.items() if k.startswith(kernel.replace("_kernel", ""))]
        return (hits[0], src) if len(hits) == 1 else (None, None)
    except (OSError, KeyError, ValueError, IndexError):
        return None, None


def profiled_mfma_util(workload_key, kernel_prefix, is_default_shape):
    """Matrix-pipe utilisation of the filter kernel (SQ_VALU_MFMA_BUSY_CYCLES over the kernel's cycles on all 1024
    SIMDs) from the newest committed rocprofv3 SQ pass of this same bench command, and the file it comes from: a
    constant of that profile, not a measurement of this run."""
    if not is_default_shape:
        return None, None
    try:
        import glob
        newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_mfma_util.json")))[-1]
        table = json.load(open(newest))[workload_key]
        hits = [v["mfma_util"] for k, v in table.items() if k.startswith(kernel_prefix)]
        return (hits[0], os.path.relpath(newest, ROOT)) if len(hits) == 1 else (None, None)
    except (OSError, KeyError, ValueError, IndexError):
        return None, None


def measured_stream_rates(torch, dev, nbytes=1 << 31, reps=10):
    """(fill GB/s, copy GB/s counting read + write) of torch's own fill_ / copy_ on a 2 GiB buffer."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = []
    for fn, moved in ((lambda: a.fill_(1), nbytes), (lambda: b.copy_(a), 2 * nbytes)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        out.append(moved * reps / (time.perf_counter() - t0) / 1e9)
    return out[0], out[1]


def synthetic_model(kind, n, seed):
    """(n, 3) float64 model points inside the [-0.4, 0.4] cube of the reference's normalised CAD models."""
    if kind == "cube":
        return np.random.RandomState(seed).uniform(-0.4, 0.4, (n, 3))
    from pose_estimation_amd import run_proj
    return run_proj.sample_surface(*run_proj.chair_mesh(), n, seed=seed)


MODEL_NAMES = {"chair": "surface samples of a box-built chair mesh", "cube": "points uniform in a solid cube"}


def run_geometry(args, torch, dist, world, rank, dev):
    """Workloads 'project' and 'disparity' (inputs resident in HBM, one GPU's share per rank)."""
    import pose_estimation_amd as pea
    from pose_estimation_amd import _abi, _device as dv, sharding
    n = args.points
    point_order_note = None
    rs = np.random.RandomState(100 + rank)
    pts_np = synthetic_model(args.model, n, 100 + rank)
    if args.workload == "zbuffer" and not args.unfused:
        args.morton = True                           # the tiled pass bins chunks of consecutive points per tile
    if args.morton:                                  # one-time model preprocessing, outside the timed region
        pts_np = pts_np[pea.run_proj.spatial_order(pts_np)]
    pts = torch.from_numpy(pts_np).to(dev)
    st = lambda: dv.stream_ptr()
    sharded = False                                  # config 4 over several ranks: the total work is fixed
    crop_one = torch.tensor([[100, 400, 300, 700]], dtype=torch.int32, device=dev)
    if args.workload == "project":
        P = args.poses
        poses = np.array([[az, el, 0.0, 3.0] for az in range(360) for el in range(-10, 40, 5)], dtype=np.float64)
        if world > 1 and P == len(poses):                # config 4: the pose grid sharded over the ranks
            sharded = True
            lo, hi = sharding.shard_range(len(poses), rank, world)
            poses = poses[lo:hi]
            P = hi - lo
        else:
            poses = np.resize(poses, (P, 4))
        cams = torch.from_numpy(pea.pt2render.camera_matrices(poses)).to(dev)
        per_pose = 1 if getattr(args, "depth_per_pose", False) else 0
        depth = torch.full((P, 540, 960) if per_pose else (540, 960), 3.0, dtype=torch.float32, device=dev)
        out = torch.empty((P, 3, n), dtype=torch.float64, device=dev)
        truth = torch.empty((P, n), dtype=torch.uint8, device=dev)

        def step():
            _abi.check(_abi.lib.ispa_project_prune(pts.data_ptr(), n, cams.data_ptr(), P, depth.data_ptr(), per_pose, 540,
                                                   960, 540, 950, 0.15, _abi.VIS_A, crop_one.data_ptr(), 0,
                                                   out.data_ptr(), truth.data_ptr(), None, None, st()))
        units, unit_name, recs = P, "pose-maps/s", 1
        alg_bytes = [24.0 * n + 29.0 * n * P]
        names = ["project_prune"]
        workload = (f"projection + depth visibility (variant A) + crop remap of a {n}-point model at {P} poses "
                    f"per GPU per step, float64 drop-in layout, "
                    + ("one depth map per pose" if per_pose else "one depth map shared by all poses"))
    elif args.workload == "corr":
        from pose_estimation_amd import correspondence as co
        bsz, views, kpts = 8, 3, 10                     # multi_view_configs/train_config.py: batch 8, 3 views
        g = torch.Generator(device=dev).manual_seed(7 + rank)

        def unit(*shape):
            x = torch.randn(*shape, device=dev, generator=g)
            return (x / x.norm(dim=1, keepdim=True)).contiguous()
        # eccv_net.Net.forward's inputs: real and template features per view, one keypoint list per view
        f_i = torch.stack([unit(bsz, 128, 56, 56) for _ in range(views)], dim=1).contiguous()
        f_j = torch.stack([unit(bsz, 128, 56, 56) for _ in range(views)], dim=1).contiguous()
        knn = np.random.RandomState(7 + rank).randint(0, 28, (views, kpts, 2))     # host list, as in the reference
        mask = (torch.rand(bsz, 1, 56, 56, device=dev, generator=g) > 0.3).float()
        layer = co.CorrelationMapLayer()
        f_real, f_tmpl0 = f_i[:, 0].contiguous(), f_j[:, 0].contiguous()

        def step():
            layer.forward_views(f_i, f_j, knn)                   # eccv_net.py:282-289, all views in one call
            co.correlation_map_acmm(f_real, f_tmpl0, mask)       # net_final_old: one dense map per pair
        units, unit_name, recs = bsz, "pairs/s", 2
        cells = 28 * 28
        alg_bytes = [(4.0 * bsz * 128 * cells + 4.0 * bsz * kpts * cells) * views, 2.0 * cells * cells * 128 * bsz]
        names = ["corr_eccv", "corr_acmm_gemm"]
        workload = (f"correlation maps at the reference's size: {views} x eccv keypoint maps (K={kpts}) + 1 x acmm "
                    f"dense map for {bsz} pairs of 128x56x56 features per GPU per step")
    elif args.workload == "zbuffer":
        P = args.poses
        poses = np.array([[az, el, 0.0, 3.0] for az in range(360) for el in range(-10, 40, 5)], dtype=np.float64)
        if world > 1 and P == len(poses):                # config 4: the pose grid sharded over the ranks
            sharded = True
            lo, hi = sharding.shard_range(len(poses), rank, world)
            poses = poses[lo:hi]
            P = hi - lo
        else:
            poses = np.resize(poses, (P, 4))
        cams = torch.from_numpy(pea.pt2render.camera_matrices(poses)).to(dev)
        truth = torch.empty((P, n), dtype=torch.uint8, device=dev)
        packed = torch.empty((P, 540, 960), dtype=torch.int64, device=dev)
        if args.unfused:
            out = torch.empty((P, 3, n), dtype=torch.float64, device=dev)

            def step():
                # project -> packed (depth | index) z-buffer by global atomics -> visibility from the packed keys
                _abi.check(_abi.lib.ispa_project(pts.data_ptr(), n, cams.data_ptr(), P, out.data_ptr(), st()))
                _abi.check(_abi.lib.ispa_zbuffer(out.data_ptr(), n, P, 540, 960, 0, packed.data_ptr(), None, None, st()))
                _abi.check(_abi.lib.ispa_zbuffer_visibility_packed(out.data_ptr(), n, P, packed.data_ptr(), 540, 960,
                                                                   0.01, truth.data_ptr(), st()))
            recs = 2
            # project: points read once + 24 B out per point-pose; z-buffer update kernel: 24 B in and one
            # 8-byte atomic per point-pose (the memset and resolve launches around it are not in its time)
            alg_bytes = [24.0 * n + 24.0 * n * P, 32.0 * n * P]
            names = ["project", "zbuffer"]
            how = "ispa_project + global-atomic ispa_zbuffer + ispa_zbuffer_visibility_packed"
        else:
            need = _abi.lib.ispa_project_zbuffer_workspace_bytes(n, P, 540, 960)
            ws = torch.empty(need, dtype=torch.uint8, device=dev)

            def step():
                _abi.check(_abi.lib.ispa_project_zbuffer(pts.data_ptr(), n, cams.data_ptr(), P, 540, 960, 0, 0.01,
                                                         packed.data_ptr(), None, None, truth.data_ptr(),
                                                         ws.data_ptr(), need, st()))
            recs = 1
            # the fused pass: the model read once, per pose the key image written once and one flag per point
            alg_bytes = [24.0 * n + (8.0 * 540 * 960 + 1.0 * n) * P]
            names = ["zbuffer_tiled"]
            how = "fused screen-tiled pass (ispa_project_zbuffer): keys in LDS, key image and flags written once"
        units, unit_name = P, "pose-maps/s"
        workload = (f"projection + packed (depth|index) z-buffer (540x960) + visibility of a {n}-point model at {P} "
                    f"poses per GPU per step; {how}")
    else:
        M = args.maps
        params = np.stack([rs.uniform(0, 360, M), rs.uniform(-10, 35, M), np.zeros(M), np.full(M, 3.0)], axis=1)
        cams1 = torch.from_numpy(pea.pt2render.camera_matrices(params)).to(dev)
        more = params.copy()
        more[:, :2] += 10.0
        cams2 = torch.from_numpy(pea.pt2render.camera_matrices(more)).to(dev)
        depth = torch.full((M, 540, 960), 3.0, dtype=torch.float32, device=dev)
        crops = crop_one.repeat(M, 1).contiguous()
        uv1 = torch.empty((M, 3, n), dtype=torch.float64, device=dev)
        uv2 = torch.empty((M, 3, n), dtype=torch.float64, device=dev)
        t1 = torch.empty((M, n), dtype=torch.uint8, device=dev)
        t2 = torch.empty((M, n), dtype=torch.uint8, device=dev)
        flow = torch.empty((M, 224, 224, 2), dtype=torch.float64, device=dev)
        owner = torch.empty((M, 224 * 224), dtype=torch.int32, device=dev)

        def prune(cams, uv, t):
            _abi.check(_abi.lib.ispa_project_prune(pts.data_ptr(), n, cams.data_ptr(), M, depth.data_ptr(), 1, 540,
                                                   960, 540, 950, 0.15, _abi.VIS_A, crops.data_ptr(), 1,
                                                   uv.data_ptr(), t.data_ptr(), None, None, st()))

        if args.unfused:
            def step():
                prune(cams1, uv1, t1)
                prune(cams2, uv2, t2)
                _abi.check(_abi.lib.ispa_flow_splat(uv1.data_ptr(), uv2.data_ptr(), t1.data_ptr(), n, M, 2, 224,
                                                    owner.data_ptr(), flow.data_ptr(), st()))
            units, unit_name, recs = M, "maps/s", 3
            alg_bytes = [24.0 * n + 29.0 * n * M, 24.0 * n + 29.0 * n * M, (33.0 * n + 802816.0) * M]
            names = ["project_prune", "project_prune", "flow_band"]
            workload = (f"make_disparity flow maps: {M} renders per GPU per step of a {n}-point model "
                        f"(2 x projection+visibility+crop, then the 224x224x2 float64 flow splat)")
        else:
            need = _abi.lib.ispa_make_disparity_workspace_bytes(n, M)
            dws = torch.empty(max(need, 4), dtype=torch.uint8, device=dev)
            # the model's points along a Hilbert curve (once per model, outside the timed region), ranks = the order
            # they were generated in: the flow maps are those of the generated order, bit for bit
            ordered = not (args.one_launch or args.morton or args.as_generated)
            po = pea.disparity.PointOrder(pts_np) if ordered else None

            def step():
                if args.one_launch:
                    _abi.check(_abi.lib.ispa_make_disparity(pts.data_ptr(), n, cams1.data_ptr(), cams2.data_ptr(), M,
                                                            depth.data_ptr(), 1, 540, 960, 540, 950, 0.15,
                                                            crops.data_ptr(), 1, 2, flow.data_ptr(), st()))
                elif po is not None:
                    _abi.check(_abi.lib.ispa_make_disparity_ordered(
                        po.points.data_ptr(), po.rank.data_ptr(), po.inverse.data_ptr(), n, cams1.data_ptr(),
                        cams2.data_ptr(), M, depth.data_ptr(), 1, 540, 960, 540, 950, 0.15, crops.data_ptr(), 1, 2,
                        flow.data_ptr(), dws.data_ptr(), need, st()))
                else:
                    _abi.check(_abi.lib.ispa_make_disparity_ws(pts.data_ptr(), n, cams1.data_ptr(), cams2.data_ptr(), M,
                                                               depth.data_ptr(), 1, 540, 960, 540, 950, 0.15,
                                                               crops.data_ptr(), 1, 2, flow.data_ptr(),
                                                               dws.data_ptr(), need, st()))
            units, unit_name, recs = M, "maps/s", 1
            if po is not None:
                point_order_note = ("processed along a hilbert curve (disparity.PointOrder), ranked as generated: the "
                                    "maps of the generated order")
            # fused: points read once per map, one depth gather per point, the flow map written
            alg_bytes = [(24.0 * n + 4.0 * n + 802816.0) * M]
            names = ["make_disparity" if args.one_launch else "make_disparity_ws"]
            workload = (f"make_disparity flow maps, fused ({'one launch, no workspace' if args.one_launch else 'with 16 B of workspace per point and map: one workgroup per map when the maps fill the chip, else per-band lists in three launches'}): "
                        f"{M} renders per GPU per step of a {n}-point "
                        f"model (projection at 2 poses, depth visibility, crop, 224x224x2 float64 flow splat)")
    for _ in range(args.warmup):
        step()
    _abi.check(_abi.lib.ispa_timing_enable(args.steps * recs))
    elapsed = sharding.timed_steps(step, lambda: None, args.steps, 0, torch.cuda.synchronize)
    ms = (ctypes.c_float * (args.steps * recs))()
    cnt = ctypes.c_int(0)
    _abi.check(_abi.lib.ispa_timing_collect(ms, args.steps * recs, ctypes.byref(cnt)))
    _abi.check(_abi.lib.ispa_timing_enable(0))
    per = [float(np.mean(ms[i:cnt.value:recs])) for i in range(recs)]
    dom = int(np.argmax(per))
    if rank != 0:
        return None
    achieved = alg_bytes[dom] / (per[dom] * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4)}
    if args.workload != "corr":
        # what this box's memory system takes when nothing but stores (fill) or loads + stores (copy) are asked of
        # it: the practical ceiling beside the nominal 8 TB/s
        fill, copy = measured_stream_rates(torch, dev)
        roof.update(measured_fill_gbs=round(fill, 1), measured_copy_gbs=round(copy, 1),
                    frac_of_fill=round(achieved / fill, 4))
    if names[dom] == "corr_acmm_gemm":                  # alg_bytes holds FLOP for this kernel
        tf = alg_bytes[dom] / (per[dom] * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_PEAK_TFLOPS["f32"], "unit": "TFLOP/s",
                "frac": round(tf / MFMA_PEAK_TFLOPS["f32"], 4)}
    traffic, traffic_src = measured_traffic(
        args.workload + ("_unfused" if args.unfused else "") + ("_depth_per_pose" if getattr(args, "depth_per_pose", False) else ""),
        {"project_prune": "project_prune2_kernel", "flow_band": "flow_band_kernel", "project": "project2_kernel",
         "zbuffer": "zbuffer_kernel", "zbuffer_tiled": "zb_tile_kernel", "corr_eccv": "corr_eccv_tile_kernel",
         "corr_acmm_gemm": "corr_acmm_gemm_kernel", "make_disparity": "make_disparity_kernel",
         "make_disparity_ws": "disparity_*"}[names[dom]],
        args.workload == "corr" or (n == 100000 and not sharded and
                                    (args.poses == 3600 if args.workload in ("project", "zbuffer") else args.maps == 256)))
    if traffic:
        # the profile's counter bytes over THIS run's kernel time: the rate HBM really delivered
        roof["hbm_frac_counter"] = round(traffic / (per[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    out_json = {
        "metric": json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"],
        "value": round(units * world * args.steps / elapsed, 2), "unit": unit_name, "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
        "dtype": "f32" if args.workload == "corr" else "f64", "data": "synthetic",
        "config": {"workload": workload, "points": n, "model": MODEL_NAMES[args.model],
                   "point_order": point_order_note or ("hilbert curve (run_proj.spatial_order)" if args.morton
                                                       else "as generated")},
        "roofline": dict(roof, traffic=traffic, traffic_source=traffic_src, kernel=names[dom],
                         kernel_ms=round(per[dom], 4), work_per_launch=alg_bytes[dom],
                         all_kernels_ms={f"{i}:{names[i]}": round(per[i], 4) for i in range(recs)}),
    }
    if not args.no_cpu_baseline and world == 1:
        out_json["cpu_baseline"] = cpu_baseline_geometry(args)
    return out_json


def disparity_step(torch, dev, maps, n, model, seed):
    """One make_disparity call for `maps` renders of an n-point model, as the `disparity` workload runs it (points
    along a Hilbert curve ranked as generated, one depth map per render): (step function, buffers to keep alive)."""
    import pose_estimation_amd as pea
    from pose_estimation_amd import _abi, _device as dv
    rs = np.random.RandomState(seed)
    pts_np = synthetic_model(model, n, seed)
    params = np.stack([rs.uniform(0, 360, maps), rs.uniform(-10, 35, maps), np.zeros(maps), np.full(maps, 3.0)], axis=1)
    cams1 = torch.from_numpy(pea.pt2render.camera_matrices(params)).to(dev)
    more = params.copy()
    more[:, :2] += 10.0
    cams2 = torch.from_numpy(pea.pt2render.camera_matrices(more)).to(dev)
    depth = torch.full((maps, 540, 960), 3.0, dtype=torch.float32, device=dev)
    crops = torch.tensor([[100, 400, 300, 700]], dtype=torch.int32, device=dev).repeat(maps, 1).contiguous()
    flow = torch.empty((maps, 224, 224, 2), dtype=torch.float64, device=dev)
    need = _abi.lib.ispa_make_disparity_workspace_bytes(n, maps)
    dws = torch.empty(max(need, 4), dtype=torch.uint8, device=dev)
    po = pea.disparity.PointOrder(pts_np)

    def step():
        _abi.check(_abi.lib.ispa_make_disparity_ordered(
            po.points.data_ptr(), po.rank.data_ptr(), po.inverse.data_ptr(), n, cams1.data_ptr(), cams2.data_ptr(),
            maps, depth.data_ptr(), 1, 540, 960, 540, 950, 0.15, crops.data_ptr(), 1, 2, flow.data_ptr(),
            dws.data_ptr(), need, dv.stream_ptr()))
    return step, (cams1, cams2, depth, crops, flow, dws, po)


def stage_records(args, torch, dist, dev):
    """The other stages of the path at their BASELINE sizes, a few steps each, for the `extra.stages` block of the
    default run (one GPU): the same measurement code as `--workload <stage>`, trimmed to the fields that say what
    ran and how close to its roofline.  Not part of `value`."""
    import copy
    out = {}
    for key in ("project", "project_depth_per_pose", "zbuffer", "disparity", "corr"):
        wl = "project" if key.startswith("project") else key
        a = copy.copy(args)
        # ten warm-up steps: the first half dozen launches over freshly allocated output buffers of several GB run
        # 10-15 % slower than the steady state (measured on the projection stage: 1.98 ms against 1.74 ms)
        a.workload, a.steps, a.warmup = wl, 10, 10
        a.poses, a.maps, a.points, a.model, a.unfused, a.one_launch, a.morton = 3600, 256, 100000, "chair", False, False, False
        a.as_generated = False
        a.depth_per_pose = key == "project_depth_per_pose"
        # a bounded CPU sample beside every geometry stage (the reference-shaped loops); the correlation stage
        # and the second projection line (same CPU work as the first) go without
        a.no_cpu_baseline = args.no_cpu_baseline or key in ("corr", "project_depth_per_pose")
        a.cpu_seconds = 4.0
        rec = run_geometry(a, torch, dist, 1, 0, dev)
        roof = rec["roofline"]
        out[key] = {"value": rec["value"], "unit": rec["unit"], "steps": a.steps, "ms_per_step": rec["ms_per_step"],
                    "dtype": rec["dtype"], "workload": rec["config"]["workload"],
                    "point_order": rec["config"]["point_order"],
                    "roofline": {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "hbm_frac_counter",
                                                      "traffic", "traffic_source", "kernel", "kernel_ms",
                                                      "frac_of_fill", "measured_fill_gbs") if k in roof}}
        if "cpu_baseline" in rec:
            out[key]["cpu_baseline"] = rec["cpu_baseline"]
        if wl == "corr":
            out[key]["note"] = ("66 KB-sized problem of five dependent launches: latency-bound at the reference's "
                                "shapes, the HBM fraction says only that no bandwidth is needed")
        torch.cuda.empty_cache()
    return out


def cpu_baseline_geometry(args):
    """The reference's algorithm shape on the host: numpy projection, per-point Python loops for the
    depth test and the splat (oracle restatements of pt2render.py:43-55 and make_disparity_syn.py:65-68)."""
    from oracle import disparity as odisp, geometry as og, zbuffer as ozb
    if args.workload == "corr":
        from oracle import correlation as ocorr
        rs = np.random.RandomState(0)
        f1 = rs.randn(8, 128, 56, 56).astype(np.float32)
        f2 = rs.randn(8, 128, 56, 56).astype(np.float32)
        knn = rs.randint(0, 28, (10, 2))
        mask = (rs.rand(8, 1, 56, 56) > 0.3).astype(np.float32)
        done, t0 = 0, time.perf_counter()
        while True:
            for _ in range(3):
                ocorr.correlation_map_eccv(f1, f2, knn)
            ocorr.correlation_map_acmm(f1, f2, mask)
            done += 1
            el = time.perf_counter() - t0
            if el >= args.cpu_seconds or done >= 50:
                break
        return {"value": round(8 * done / el, 3), "unit": "pairs/s", "cores": os.cpu_count() or 1, "kind": "port",
                "cpu_model": cpu_model(),
                "sample": f"{done} steps (3 x eccv K=10 + 1 x acmm, B=8) in {el:.1f} s; numpy einsum restatement "
                          f"(oracle.correlation), which contracts only the K needed rows unlike the reference's "
                          f"5-D broadcast"}
    n = args.points
    rs = np.random.RandomState(0)
    pts = synthetic_model(args.model, n, 0)
    depth = np.full((540, 960), 3.0, np.float32)
    crop = [100, 400, 300, 700]
    done, t0 = 0, time.perf_counter()
    while True:
        params = [30.0 + done, 10.0, 0.0, 3.0]
        r1 = og.get_3d_to_2d(pts, params)
        if args.workload == "zbuffer":
            zdepth = ozb.zbuffer(r1, 540, 960)[2]
            ozb.zbuffer_visibility(r1, zdepth, 0.01)
            done += 1
            el = time.perf_counter() - t0
            if el >= args.cpu_seconds or done >= 50:
                break
            continue
        t1 = og.depth_visibility_literal(r1, depth, "A")
        c1, _ = og.crop_visibility_array(r1, crop)
        if args.workload == "disparity":
            r2 = og.get_3d_to_2d(pts, [params[0] + 10, params[1] + 10, 0.0, 3.0])
            og.depth_visibility_literal(r2, depth, "A")
            og.crop_visibility_array(r2, crop)
            keep = t1 & c1
            odisp.splat_flow_literal(r1[:, keep], (r2 - r1)[:, keep])
        done += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or done >= 50:
            break
    if args.workload == "zbuffer":
        return {"value": round(done / el, 3), "unit": "pose-maps/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
                "sample": f"{done} poses of {n} points in {el:.1f} s; numpy projection + np.minimum.at z-buffer "
                          f"(oracle.zbuffer; this mode has no counterpart in the reference)"}
    what = "make_disparity images" if args.workload == "disparity" else "project+prune poses"
    return {"value": round(done / el, 3), "unit": "maps/s" if args.workload == "disparity" else "pose-maps/s",
            "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{done} {what} of {n} points in {el:.1f} s; numpy projection + per-point Python loops as in "
                      f"the reference (oracle.geometry.depth_visibility_literal, oracle.disparity.splat_flow_literal)"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    import pose_estimation_amd as pea           # raises if libispa_hip.so is missing
    from pose_estimation_amd import _abi, sharding

    dev = torch.device("cuda", torch.cuda.current_device())
    if args.workload != "nn":
        rec = run_geometry(args, torch, dist, world, rank, dev)
        if rec is not None:
            print(json.dumps(rec))
        if world > 1:
            dist.destroy_process_group()
        return
    b, c, g = args.pairs, args.channels, args.grid
    n = g * g
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    def feats():
        x = torch.randn(b, c, g, g, device=dev, generator=gen)
        return (x / x.norm(dim=1, keepdim=True)).to(tdt).contiguous()
    q, r = feats(), feats()
    if args.matches:
        perm = torch.stack([torch.randperm(n, device=dev, generator=gen) for _ in range(b)])
        q = torch.gather(r.float().view(b, c, n), 2, perm[:, None, :].expand(-1, c, -1)).view(b, c, g, g)
        q = q + 0.02 * torch.randn(b, c, g, g, device=dev, generator=gen)
        q = (q / q.norm(dim=1, keepdim=True)).to(tdt).contiguous()
    # gloo cannot all-gather device tensors: the rehearsal backend runs without the result exchange
    gather = world > 1 and not args.no_gather and args.backend == "nccl"
    filt = args.filter if args.dtype == "f32" else "native"

    pending = []                                    # result gathers of the previous step, still in flight

    def step():
        dist_, ind_ = pea.correspondence.dense_nn(q, r, args.k, filter=filt)
        if gather:
            # the records of step i travel over xGMI while step i + 1 computes; every gather is
            # completed inside the timed region (here or in drain())
            for p_ in pending:
                p_.wait()
            pending[:] = [sharding.allgather_results_async(ind_, b * world),
                          sharding.allgather_results_async(dist_, b * world)]
        return dist_, ind_

    def drain():
        for p_ in pending:
            p_.wait()
        pending.clear()

    for _ in range(args.warmup):
        step()
    drain()
    _abi.check(_abi.lib.ispa_timing_enable(args.steps))
    elapsed = sharding.timed_steps(step, drain, args.steps, 0, torch.cuda.synchronize)

    # per-launch duration of the filter kernel, HIP events on the launch stream inside the timed region
    ms = (ctypes.c_float * args.steps)()
    cnt = ctypes.c_int(0)
    _abi.check(_abi.lib.ispa_timing_collect(ms, args.steps, ctypes.byref(cnt)))
    _abi.check(_abi.lib.ispa_timing_enable(0))
    filt_ms = float(np.mean(ms[:cnt.value])) if cnt.value else float("nan")

    # Not part of `value`: the same batch through the bfloat16 path (inputs rounded to bfloat16, results
    # exact for those inputs), reported beside the float32 headline.
    extra = {}

    def side_run(fn, n_steps=10):
        """(elapsed seconds over ranks, mean filter-kernel ms) of n_steps calls of fn, after five warm-up calls."""
        for _ in range(5):
            fn()
        _abi.check(_abi.lib.ispa_timing_enable(n_steps))
        el = sharding.timed_steps(fn, lambda: None, n_steps, 0, torch.cuda.synchronize)
        buf = (ctypes.c_float * n_steps)()
        got = ctypes.c_int(0)
        _abi.check(_abi.lib.ispa_timing_collect(buf, n_steps, ctypes.byref(got)))
        _abi.check(_abi.lib.ispa_timing_enable(0))
        return el, (float(np.mean(buf[:got.value])) if got.value else float("nan"))

    if args.dtype == "f32" and not args.no_extra:
        n16 = 20
        q16, r16 = q.to(torch.bfloat16), r.to(torch.bfloat16)
        el16, k16 = side_run(lambda: pea.correspondence.dense_nn(q16, r16, args.k), n16)
        tf16 = 2.0 * n * n * c * b / (k16 * 1e-3) / 1e12
        extra["bf16"] = {"value": round(b * world * n16 / el16, 2), "unit": "maps/s", "steps": n16,
                         "ms_per_step": round(el16 / n16 * 1e3, 4), "result_allgather": False,
                         "roofline": {"bound": "mfma", "achieved": round(tf16, 2), "peak": MFMA_PEAK_TFLOPS["bf16"],
                                      "unit": "TFLOP/s", "frac": round(tf16 / MFMA_PEAK_TFLOPS["bf16"], 4),
                                      "kernel": "nn_filter", "kernel_ms": round(k16, 4)}}
        del q16, r16
        if args.filter == "native" and args.k == 1:
            els, ks = side_run(lambda: pea.correspondence.dense_nn(q, r, 1, filter="split"), n16)
            extra["f32_inputs_split_bf16_filter"] = {
                "value": round(b * world * n16 / els, 2), "unit": "maps/s", "steps": n16,
                "ms_per_step": round(els / n16 * 1e3, 4), "result_allgather": False,
                "note": "same float32 inputs and the same exact results as the headline; the filter pass contracts "
                        "a bfloat16 hi/lo split on bfloat16 MFMAs (3 MFMAs per tile and k-step)",
                "kernel": "nn_filter", "kernel_ms": round(ks, 4)}
        if not args.matches:
            # the same batch with queries that have close matches (noisy copies of template cells): every returned
            # distance then takes the float64 refinement, which random features never exercise
            perm = torch.stack([torch.randperm(n, device=dev, generator=gen) for _ in range(b)])
            qm = torch.gather(r.view(b, c, n), 2, perm[:, None, :].expand(-1, c, -1)).view(b, c, g, g)
            qm = qm + 0.02 * torch.randn(b, c, g, g, device=dev, generator=gen)
            qm = (qm / qm.norm(dim=1, keepdim=True)).contiguous()
            elm, km = side_run(lambda: pea.correspondence.dense_nn(qm, r, args.k, filter=filt), n16)
            extra["f32_close_matches"] = {
                "value": round(b * world * n16 / elm, 2), "unit": "maps/s", "steps": n16,
                "ms_per_step": round(elm / n16 * 1e3, 4), "result_allgather": False,
                "note": "queries = noisy copies of template cells: all distances re-scored in float64 (nn_refine_kernel)",
                "kernel": "nn_filter", "kernel_ms": round(km, 4)}
            del qm, perm

    default_shape = (b, g, c, args.k, args.dtype, filt) == (512, 64, 128, 1, "f32", "native") and not args.matches
    if world == 1 and default_shape and not args.no_extra:
        # BASELINE config 2 literally: ONE 64x64x128 float32 pair per call (a latency case: a chain of five short
        # dependent launches), pipelined on one stream and with a host wait after every call
        q1, r1 = q[:1].contiguous(), r[:1].contiguous()
        el1, k1 = side_run(lambda: pea.correspondence.dense_nn(q1, r1, 1), 200)
        for _ in range(20):
            pea.correspondence.dense_nn(q1, r1, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            pea.correspondence.dense_nn(q1, r1, 1)
            torch.cuda.synchronize()
        waited = (time.perf_counter() - t0) / 200
        extra["single_pair_f32"] = {
            "value": round(200 / el1, 1), "unit": "pairs/s", "calls": 200, "us_per_call_pipelined": round(el1 / 200 * 1e6, 1),
            "us_per_call_waited": round(waited * 1e6, 1), "filter_kernel_us": round(k1 * 1e3, 1),
            "filter_tflops": round(2.0 * n * n * c / (k1 * 1e-3) / 1e12, 1),
            "workload": f"one {g}x{g}x{c} float32 pair per call (BASELINE config 2)"}
        del q1, r1
        # The metric names correspondence AND disparity maps: per image pair one dense NN correspondence and one
        # make_disparity flow map (100k-point model) in the same timed region, back to back on one stream.
        disp, keep = disparity_step(torch, dev, b, 100000, "chair", 300)
        q16, r16 = q.to(torch.bfloat16), r.to(torch.bfloat16)
        combined = {"unit": "pairs/s", "steps": 10,
                    "workload": f"per pair one dense NN correspondence ({g}x{g}x{c}, k=1) + one make_disparity flow "
                                f"map (224x224x2 float64, 100000-point model), {b} pairs per step"}
        for tag, fn in (("f32", lambda: (pea.correspondence.dense_nn(q, r, 1), disp())),
                        ("bf16", lambda: (pea.correspondence.dense_nn(q16, r16, 1), disp()))):
            elc, _ = side_run(fn, 10)
            combined[tag] = {"value": round(b * 10 / elc, 2), "ms_per_step": round(elc / 10 * 1e3, 4)}
        extra["combined"] = combined
        del q16, r16, disp, keep
        del q, r
        torch.cuda.empty_cache()
        extra["stages"] = stage_records(args, torch, dist, dev)

    if rank == 0:
        pairs_total = b * world * args.steps
        flop_per_pair = 2.0 * n * n * c                          # GEMM term of the distance stage
        achieved = flop_per_pair * b / (filt_ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS["bf16" if filt == "split" else args.dtype]
        traffic, traffic_src = measured_traffic(
            "nn___dtype_" + args.dtype + ("___filter_split" if filt == "split" else ""),
            "nn_filter_split_kernel" if filt == "split" else "nn_filter_%s_kernel" % args.dtype,
            (b, g, c, args.k) == (512, 64, 128, 1) and not args.matches)
        util, util_src = profiled_mfma_util(
            "nn___dtype_" + args.dtype + ("___filter_split" if filt == "split" else ""), "nn_filter",
            (b, g, c, args.k) == (512, 64, 128, 1) and not args.matches)
        out = {
            "metric": json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"],
            "value": round(pairs_total / elapsed, 2),
            "unit": "maps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16x3" if filt == "split" else args.dtype,
            "data": "synthetic",
            "config": {"workload": f"dense NN correspondence, {g}x{g} grid x {c}-d features, k={args.k}, "
                                   f"{b} pairs per GPU per step ("
                                   + ("float32 inputs, split-bfloat16 MFMA filter" if filt == "split"
                                      else f"{args.dtype} MFMA filter") + " + float64 exact pass)",
                       "pairs_per_gpu": b, "grid": g, "channels": c, "k": args.k,
                       "features": "queries = noisy copies of template cells" if args.matches else "independent random unit vectors",
                       "result_allgather": bool(gather)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "nn_filter", "kernel_ms": round(filt_ms, 4),
                         "flop_per_launch": flop_per_pair * b,
                         "mfma_util_profiled": util, "mfma_util_source": util_src},
        }
        if extra:
            out["extra"] = extra
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(g, c, args.k, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
