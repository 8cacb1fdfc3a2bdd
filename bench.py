#!/usr/bin/env python3
"""Headline benchmark: BDPT throughput (Mpaths/s) of the HIP backend.

  python bench.py --gpus N --steps K --warmup W

Default workload (BASELINE.json configs[3], the configuration the metric "at 1/2/4/8 GPUs" is quoted on):
C4, the Cornell box with 2 M-triangle meshes at 1920x1080. One "step" = one ubdpt_enqueue_passes call of
--passes-per-step progressive passes (default 16, 33 M pixel-samples); the default 16 steps render the
configuration's 256 spp on one GPU (about 3 s). Scene, tree and sum buffer are resident in HBM before the
timed region. --config C1|C2|C3|C5 selects the other configurations.

N > 1 (launched by torch.distributed.run, one rank per GPU): rank r renders passes r, r+N, r+2N, ...
(pass_stride = N, no data-path collective while rendering); after the last step the per-GPU sum
buffers are combined by one RCCL reduce. Work per GPU is fixed => "scaling": "weak".

The JSON line also carries
  roofline     — algorithmic HBM bytes of the dominant kernel / its HIP-event time / 8 TB/s
  cpu_baseline — the CPU oracle (restatement of the reference renderer) timed on this host on a bounded
                 sample of the same workload: the reference's own scans for the box scenes, the oracle's
                 per-mesh tree (Mode B) for the mesh scenes, where the reference's scan of every face is
                 not runnable (its cost per ray is measured on a few rays and reported beside it)
  other_configs — short runs of the remaining GPU configurations (N = 1 only, --no-other-configs skips them)
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bidirectional-pathtracing_amd")
capi, scenes, multigpu = pkg.capi, pkg.scenes, pkg.multigpu

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
WORKLOADS = {
    "C1": "Cornell box, two Lambertian spheres",
    "C2": "Cornell box, dielectric + perfect-mirror spheres",
    "C3": "Cornell box with 100 k-triangle textured meshes",
    "C4": "Cornell box with 2 M-triangle meshes",
    "C5": "Cornell box with 100 k-triangle meshes, Lambertian / glossy / mirror / dielectric",
}
MESH_CONFIGS = ("C3", "C4", "C5")


def host_threads():
    """Threads for the CPU baseline: the CPU share the box grants this job (OMP_NUM_THREADS, which the GPU
    boxes set to the share of one GPU), otherwise every CPU this process may run on."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        avail = os.cpu_count() or 1
    share = int(os.environ.get("OMP_NUM_THREADS", "0"))
    return max(1, min(share, avail) if share > 0 else avail)


def algorithmic_bytes(k, nodes, prims):
    """Algorithmic bytes of one kernel's counted work: closest-hit ray 32 B in + 32 B out, shadow ray
    32 B in + 4 B out, 64 B per tree node, 48 B per primitive tested, 96 B per path vertex written
    (subpath kernels) or read (connection kernel), 24 B per framebuffer read-modify-write.
    nodes / prims: the traversal counts to price (the CPU oracle's binary tree, the device's own tree, or 0 for the
    compulsory-only figure)."""
    return (64.0 * k["closest_hit_rays"] + 36.0 * k["shadow_rays"] + 64.0 * nodes
            + 48.0 * prims + 96.0 * k["vertices"] + 24.0 * k["fb_updates"])


def measure_work(scene, params, device, passes, seed, accel, pipeline=0, ref_tree=None):
    """Counts the work of `passes` passes with the counting build of the kernels (deterministic).
    ref_tree = {"nodes_closest", "prims_closest", "nodes_shadow", "prims_shadow"} per ray from the CPU oracle's
    tree, or None: then the device's own counts are used."""
    opt = capi.Options()
    opt.collect_stats = 1
    opt.accel = accel
    opt.pipeline = pipeline
    with capi.Context(scene, params, opt, device) as ctx:
        ctx.render_passes(0, passes, seed)
        s = ctx.stats()
    ps = float(s["pixel_samples"])
    per_kernel, per_kernel_device, per_kernel_compulsory = {}, {}, {}
    ev = s["eye_vertices"]; lv = s["light_vertices"]
    # vertex records: the generation kernel writes the lens and the emitter vertex of every
    # pixel-sample, the bounce / tail / per-path kernels the rest (split by their share of closest-hit
    # rays); the connection stage reads every vertex once
    # k_wf_shade pushes the vertices of the hits that k_wf_trace found, so its share is trace's rays
    by_name = {k["name"]: k for k in s["kernels"]}
    writer_rays = {}
    for name in ("k_wf_tail", "k_eye", "k_light"):
        if name in by_name:
            writer_rays[name] = float(by_name[name]["closest_hit_rays"])
    if "k_wf_shade" in by_name and "k_wf_trace" in by_name:
        writer_rays["k_wf_shade"] = float(by_name["k_wf_trace"]["closest_hit_rays"])
    wrays = sum(writer_rays.values()) or 1.0
    has_gen = "k_wf_gen" in by_name
    rest = (ev + lv - 2 * ps) if has_gen else (ev + lv)
    # the connection stage reads every vertex once; with the split pair kernels that read is charged
    # to the evaluation kernel (the visibility kernel is charged its shadow rays and their traversal)
    readers = ("k_connect", "k_pair_eval")
    for k in s["kernels"]:
        kk = dict(k)
        if k["name"] in readers:
            kk["vertices"] = ev + lv
        elif k["name"] == "k_wf_gen":
            kk["vertices"] = 2 * ps
        elif k["name"] in writer_rays:
            kk["vertices"] = rest * writer_rays[k["name"]] / wrays
        else:
            kk["vertices"] = 0
        dev_nodes, dev_prims = k["nodes_visited"], k["prims_tested"]
        if ref_tree is not None:
            ref_nodes = k["closest_hit_rays"] * ref_tree["nodes_closest"] + k["shadow_rays"] * ref_tree["nodes_shadow"]
            ref_prims = k["closest_hit_rays"] * ref_tree["prims_closest"] + k["shadow_rays"] * ref_tree["prims_shadow"]
        else:
            ref_nodes, ref_prims = dev_nodes, dev_prims
        per_kernel[k["name"]] = algorithmic_bytes(kk, ref_nodes, ref_prims) / ps
        per_kernel_device[k["name"]] = algorithmic_bytes(kk, dev_nodes, dev_prims) / ps
        per_kernel_compulsory[k["name"]] = algorithmic_bytes(kk, 0.0, 0.0) / ps
    rays = max(1.0, s["closest_hit_rays"] + s["shadow_rays"])
    return {
        "bytes_per_pixel_sample": per_kernel,                        # traversal priced with the CPU oracle's tree (mesh scenes)
        "bytes_per_pixel_sample_device_counts": per_kernel_device,   # traversal priced with the device tree's own node / primitive visits
        "bytes_per_pixel_sample_compulsory": per_kernel_compulsory,  # traversal terms dropped
        "rays_per_pixel_sample": (s["closest_hit_rays"] + s["shadow_rays"]) / ps,
        "closest_per_pixel_sample": s["closest_hit_rays"] / ps,
        "shadow_per_pixel_sample": s["shadow_rays"] / ps,
        "nodes_per_ray": s["bvh_nodes_visited"] / rays,
        "prims_per_ray": s["prims_tested"] / rays,
        "lanes_per_node_step": {k["name"]: k["nodes_visited"] / k["wave_node_steps"] for k in s["kernels"] if k["wave_node_steps"]},
        "max_stack_entries": max([k["max_stack_entries"] for k in s["kernels"]] + [0]),
        "eye_vertices": ev / ps, "light_vertices": lv / ps, "splats": s["splats"] / ps,
    }


def cpu_baseline(config, scene, params, seed, budget_s=12.0):
    """Times the CPU oracle on a bounded sample of the same workload. Box scenes (C1, C2): the reference's own
    scans at the full resolution. Mesh scenes: the oracle's per-mesh tree (Mode B) at a quarter of the resolution
    in each direction (same scene, same camera: the same mix of paths per pixel-sample), and the cost of the
    reference's own scan per ray, measured on a few rays."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    cores = host_threads()
    mesh = config in MESH_CONFIGS
    W, H = int(params.img_res_x), int(params.img_res_y)
    sp = params
    if mesh:
        sp = scenes.make_params(max(1, W // 4), max(1, H // 4), int(params.pixel_subdiv), int(params.lens_subdiv),
                                int(params.min_depth), float(params.focus_plane_distance), float(params.lens_size))
    orc = oracle_py.Oracle(scene, sp, 32)
    t = time.perf_counter()
    if mesh:
        orc.set_accel(1)
    build_s = time.perf_counter() - t
    npix = int(sp.img_res_x * sp.img_res_y)
    t = time.perf_counter()
    orc.render_passes(0, 1, seed, num_threads=cores)   # warm-up pass, excluded
    t1 = time.perf_counter() - t
    reps = 3
    n = int(max(1, min(256, budget_s / reps / max(t1, 1e-3))))
    times = []
    for r in range(reps):   # three repetitions over different passes of the same workload; the median is reported
        t = time.perf_counter()
        img, cn = orc.render_passes(1 + r * n, n, seed, num_threads=cores, counters=True)
        times.append(time.perf_counter() - t)
    dt = float(np.median(times))
    rays = cn["closest_hit_rays"] + cn["shadow_rays"]
    out = {
        "value": npix * n / dt / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
        "host_cpus": os.cpu_count(), "repetitions": reps, "seconds_per_repetition": [round(x, 3) for x in times],
        "sample": (f"median of {reps} repetitions of {n} passes of the {config} scene at {int(sp.img_res_x)}x{int(sp.img_res_y)} "
                   f"({npix * n} pixel-samples each, {dt:.1f} s), fp64 restatement of the reference, "
                   + ("per-mesh bounding-volume hierarchy (Mode B; the reference's scan of every face is not runnable at this size)"
                      if mesh else "linear scans in the reference's order")
                   + f", {cores} threads = the CPU share of this job (host has {os.cpu_count()} CPUs)"),
        "mrays_per_s": rays / dt / 1e6,
        "rays_per_pixel_sample": rays / max(1, cn["pixel_samples"]),
    }
    ref_tree = None
    if mesh:
        sh = max(1, cn["shadow_rays"]); cl = max(1, cn["closest_hit_rays"])
        ref_tree = {"nodes_closest": cn["nodes_visited_closest"] / cl, "prims_closest": cn["prims_tested_closest"] / cl,
                    "nodes_shadow": (cn["nodes_visited"] - cn["nodes_visited_closest"]) / sh,
                    "prims_shadow": (cn["prims_tested"] - cn["prims_tested_closest"]) / sh}
        out["tree_build_s"] = build_s
        out["tree_per_ray"] = ref_tree
        # Mode A: the reference's algorithm itself (every face of every mesh whose bounding sphere is hit), a few rays
        # (half of the rays start at the camera and aim into the box, half start inside it: the mix of the parity tests)
        rng = np.random.default_rng(42)
        k = 32 if scene.num_faces > 500_000 else 128
        rays_a = np.zeros(k, capi.RAY_DTYPE)
        cam = np.array([scene.desc.camera.pos[0], scene.desc.camera.pos[1], scene.desc.camera.pos[2]])
        rays_a["origin"][: k // 2] = cam
        v = rng.uniform(-1.0, 1.0, (k // 2, 3)) - cam
        rays_a["dir"][: k // 2] = v / np.linalg.norm(v, axis=1, keepdims=True)
        rays_a["origin"][k // 2:] = rng.uniform(-0.999, 0.999, (k - k // 2, 3))
        v = rng.normal(size=(k - k // 2, 3))
        rays_a["dir"][k // 2:] = v / np.linalg.norm(v, axis=1, keepdims=True)
        lin = oracle_py.Oracle(scene, sp, 32)
        t = time.perf_counter()
        lin.intersect_batch(rays_a)
        ta = (time.perf_counter() - t) / k
        hours_per_pass = ta * out["rays_per_pixel_sample"] * W * H / cores / 3600.0
        out["reference_scan"] = {
            "seconds_per_ray_one_thread": ta, "rays_timed": k,
            "note": (f"the reference's own algorithm (uapp/mesh.cpp:123-137: every face of every mesh whose bounding sphere "
                     f"the ray meets) costs {ta * 1e3:.2f} ms per ray on one thread here; one pass of the {config} workload at "
                     f"{W}x{H} would take about {hours_per_pass:.2f} h on {cores} threads: not runnable")}
    return out, orc, ref_tree


def isolated_totals(scene, params, device, args, batch_ps, npix):
    """Total HIP-event time per kernel of three batches rendered on one batch lane (nothing else on the device)."""
    opt = capi.Options()
    opt.accel = args.accel
    opt.pipeline = args.pipeline
    opt.max_path_vertices = args.max_path_vertices
    opt.dense_vertex_slots = args.dense_vertex_slots
    opt.wavefront_depth = args.wavefront_depth
    opt.streams = 1
    passes = max(1, int(batch_ps // npix))
    opt.passes_in_flight = passes
    with capi.Context(scene, params, opt, device) as c1:
        c1.render_passes(1 << 41, passes, args.seed)
        c1.reset_stats()
        c1.render_passes((1 << 41) + passes, 3 * passes, args.seed)
        return {k["name"]: k["ms"] for k in c1.stats()["kernels"]}


def timed_run(ctx, args, steps, pps, rank, ngpu, dist, sum_dev, device):
    import torch

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # warm-up steps use passes beyond the measured range, then the sum is cleared
    base = 1 << 40
    for w in range(args.warmup):
        ctx.render_passes(base + (w * pps * ngpu) + rank, pps, args.seed, pass_stride=ngpu)
    if dist is not None and args.warmup:
        ctx.read_sum_device(sum_dev.data_ptr())
        multigpu.reduce_sum(sum_dev, dist)
    ctx.upload_sum(None)
    ctx.reset_stats()

    sync_all()
    t0 = time.perf_counter()
    per_step = pps * ngpu if args.scaling == "weak" else pps   # passes of one step over all GPUs
    for s in range(steps):
        # step s covers passes [s * per_step, (s + 1) * per_step); this rank takes every ngpu-th of them
        first, count, stride = multigpu.partition_passes(s * per_step, per_step, rank, ngpu)
        if count:
            ctx.enqueue_passes(first, count, args.seed, pass_stride=stride)
    ctx.read_sum_device(sum_dev.data_ptr())  # waits for the enqueued passes
    multigpu.reduce_sum(sum_dev, dist)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{device}" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed


def short_run(config, device, seed, target_s=1.0):
    """A short single-GPU run of another configuration: Mpaths/s with the sum resident on the device."""
    scene, params, _ = scenes.make_config(config)
    npix = int(params.img_res_x * params.img_res_y)
    pps = max(1, min(64, (1 << 23) // npix))
    with capi.Context(scene, params, None, device) as ctx:
        ctx.render_passes(1 << 40, pps, seed)
        ctx.upload_sum(None)
        t = time.perf_counter()
        ctx.render_passes(0, pps, seed)
        t1 = time.perf_counter() - t
        steps = int(max(1, min(64, target_s / max(t1, 1e-4))))
        ctx.upload_sum(None)
        t = time.perf_counter()
        for s in range(steps):
            ctx.enqueue_passes(s * pps, pps, seed)
        ctx.synchronize()
        dt = time.perf_counter() - t
        img = ctx.read_sum()
    return {"workload": f"{config}: {WORKLOADS[config]}, {int(params.img_res_x)}x{int(params.img_res_y)}",
            "value": npix * pps * steps / dt / 1e6, "unit": "Mpaths/s", "spp": pps * steps, "seconds": dt,
            "finite": bool(np.isfinite(img).all()), "mean_radiance": [float(x) for x in img.reshape(-1, 3).mean(0) / (pps * steps)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--passes-per-step", type=int, default=0, help="0 = 16 for C4 (256 spp in the default 16 steps), else about 33 M pixel-samples")
    ap.add_argument("--config", default="C4")
    ap.add_argument("--accel", type=int, default=0)
    ap.add_argument("--pipeline", type=int, default=0)
    ap.add_argument("--passes-in-flight", type=int, default=0)
    ap.add_argument("--max-path-vertices", type=int, default=0, help="cap on the vertices of a subpath (0 = the library's default, 32)")
    ap.add_argument("--dense-vertex-slots", type=int, default=0, help="vertex slots stored for every subpath (0 = the library's default, 16)")
    ap.add_argument("--wavefront-depth", type=int, default=0, help="bounces run as wavefront launches (0 = default)")
    ap.add_argument("--streams", type=int, default=0, help="batches in flight on separate streams (0 = default)")
    ap.add_argument("--batch-pixel-samples", type=int, default=0, help="sub-pass batches of this many pixels (0 = whole passes)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every GPU renders steps x passes-per-step passes (default); strong: the N GPUs share them "
                         "(C4 as BASELINE.json names it: 256 spp in total on 1 / 2 / 4 / 8 GPUs = the default 16 steps x 16 passes)")
    ap.add_argument("--seed", type=lambda x: int(x, 0), default=0x5EED)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal of the N>1 path)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank renders on GPU 0")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
        sys.exit(2)
    ngpu = world

    import torch
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP backend has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    dist = None
    if ngpu > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dev = 0 if args.single_device else local_rank
        torch.cuda.set_device(dev)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.dist_backend)
    device = (0 if args.single_device else local_rank) if ngpu > 1 else 0
    torch.cuda.set_device(device)

    scene, params, cfg_spp = scenes.make_config(args.config)
    W, H = int(params.img_res_x), int(params.img_res_y)
    npix = W * H
    pps = args.passes_per_step or max(1, min(64, round((1 << 25) / npix)))

    # CPU baseline first (rank 0, N = 1 only): the mesh scenes take their algorithmic node counts from its tree
    cb = orc = ref_tree = None
    if rank == 0 and ngpu == 1 and not args.no_cpu_baseline:
        cb, orc, ref_tree = cpu_baseline(args.config, scene, params, args.seed)
    work = None
    if rank == 0:
        work = measure_work(scene, params, device, 4, args.seed, args.accel, args.pipeline, ref_tree)

    opt = capi.Options()
    opt.accel = args.accel
    opt.pipeline = args.pipeline
    opt.passes_in_flight = args.passes_in_flight
    opt.max_path_vertices = args.max_path_vertices
    opt.dense_vertex_slots = args.dense_vertex_slots
    opt.wavefront_depth = args.wavefront_depth
    opt.streams = args.streams
    opt.batch_pixel_samples = args.batch_pixel_samples
    ctx = capi.Context(scene, params, opt, device)
    sum_dev = torch.zeros((H, W, 3), dtype=torch.float64, device=f"cuda:{device}")

    elapsed = timed_run(ctx, args, args.steps, pps, rank, ngpu, dist, sum_dev, device)

    if rank == 0:
        total_passes = args.steps * pps * (ngpu if args.scaling == "weak" else 1)
        pixel_samples = total_passes * npix
        value = pixel_samples / elapsed / 1e6
        st = ctx.stats()
        kernels = {k["name"]: k for k in st["kernels"]}
        batch_ps, lanes, wf_depth = ctx.batch_info()
        if ngpu == 1:
            ctx.close()   # its batch buffers hold most of the device memory; the one-lane pass below needs it
        ps_gpu = pixel_samples / ngpu   # pixel-samples this GPU rendered in the timed region
        # measured fabric bytes per launch: only from a counter profile of THIS configuration (profiles/traffic.json),
        # taken with batches of `_batch_pixel_samples` pixel-samples
        traffic_cfg = {}
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                traffic_cfg = json.load(open(tf)).get(args.config, {})
            except Exception:
                traffic_cfg = {}
        tr_batch = float(traffic_cfg.get("_batch_pixel_samples", 0) or 0)
        batches = kernels["k_wf_gen"]["launches"] if "k_wf_gen" in kernels else 0   # one generation launch per batch

        def kernel_line(k):
            """Roofline figures of one kernel: algorithmic bytes of a launch / mean HIP-event duration of a launch."""
            name, launches = k["name"], max(1, k["launches"])
            ms = k["ms"] / launches
            out_k = {"kernel": name, "ms_total": k["ms"], "ms_per_launch": ms, "launches": k["launches"]}
            for key, field in (("frac", "bytes_per_pixel_sample"), ("frac_device_counts", "bytes_per_pixel_sample_device_counts"),
                               ("frac_compulsory", "bytes_per_pixel_sample_compulsory")):
                b = work[field].get(name, 0.0) * ps_gpu / launches
                out_k[key] = (b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else 0.0
                if key == "frac":
                    out_k["algorithmic_bytes_per_launch"] = b
            t = traffic_cfg.get(name)
            if t is not None and tr_batch > 0 and batches > 0:
                per_batch = k["launches"] / batches                      # launches of this kernel per batch
                scale = batch_ps / tr_batch                              # batch of this run against the profiled one
                out_k["traffic"] = t * scale                             # counter bytes per launch, at this run's batch size
                alg = work["bytes_per_pixel_sample"].get(name, 0.0) * tr_batch / per_batch
                out_k["traffic_over_algorithmic"] = t / alg if alg > 0 else None
                out_k["frac_traffic"] = (t * scale / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else 0.0
            return out_k

        lines = sorted((kernel_line(k) for k in kernels.values() if work["bytes_per_pixel_sample"].get(k["name"])),
                       key=lambda x: -x["ms_total"])
        # Which kernel dominates: under two batch lanes a launch's duration includes the time its workgroups wait for
        # room beside the other lane's kernels, so the totals above say who waited, not who used the machine (the
        # order of the first two changes from run to run). The kernel with the largest total in a short pass on ONE
        # lane (same batch size, kernels alone on the device) is taken; its figures are those of the timed region.
        alone = isolated_totals(scene, params, device, args, batch_ps, npix) if ngpu == 1 else {}
        for l in lines:
            l["ms_total_one_lane_pass"] = alone.get(l["kernel"])
        dom = max(lines, key=lambda l: (alone.get(l["kernel"]) or 0.0)) if alone else lines[0]
        dom_ms, dom_bytes = dom["ms_per_launch"], dom["algorithmic_bytes_per_launch"]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = dom.get("traffic")
        # whole pipeline: bytes of every kernel per pixel-sample x pixel-samples per second
        pps_rate = ps_gpu / elapsed
        pipeline = {
            "algorithmic_gbs": sum(work["bytes_per_pixel_sample"].values()) * pps_rate / 1e9,
            "device_counts_gbs": sum(work["bytes_per_pixel_sample_device_counts"].values()) * pps_rate / 1e9,
            "compulsory_gbs": sum(work["bytes_per_pixel_sample_compulsory"].values()) * pps_rate / 1e9,
        }
        if tr_batch > 0 and batches > 0:
            tot = sum(traffic_cfg[n] * (k["launches"] / batches) for n, k in kernels.items() if n in traffic_cfg)
            pipeline["hbm_gbs"] = tot / tr_batch * pps_rate / 1e9          # sum of counter bytes per pixel-sample x rate
            pipeline["hbm_frac_of_peak"] = pipeline["hbm_gbs"] / HBM_PEAK_GBS
            pipeline["counter_bytes_per_pixel_sample"] = tot / tr_batch
        img = sum_dev.cpu().numpy()
        out = {
            "metric": "bdpt_mpaths_per_s", "value": value, "unit": "Mpaths/s", "n_gpus": ngpu, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {WORKLOADS.get(args.config, 'synthetic scene')}, {W}x{H}",
                       "triangles": int(scene.num_faces), "passes_per_step": pps, "spp_total": total_passes,
                       "passes_per_gpu": total_passes / ngpu,
                       "accel": "lbvh" if args.accel == 0 else "linear", "pipeline": args.pipeline,
                       "passes_in_flight": args.passes_in_flight or "auto", "max_path_vertices": args.max_path_vertices or 32, "streams": args.streams or "default",
                       "wavefront_depth": args.wavefront_depth or "default", "partition": f"passes round-robin over {ngpu} GPU(s)"},
            "timed_seconds": elapsed,
            "mrays_per_s": value * work["rays_per_pixel_sample"],
            "rays_per_pixel_sample": work["rays_per_pixel_sample"],
            "work_per_pixel_sample": {k: work[k] for k in ("closest_per_pixel_sample", "shadow_per_pixel_sample",
                                                           "nodes_per_ray", "prims_per_ray", "eye_vertices",
                                                           "light_vertices", "splats", "lanes_per_node_step",
                                                           "max_stack_entries")},
            "mean_radiance": [float(x) for x in (img.reshape(-1, 3).mean(0) / total_passes)],
            "finite": bool(np.isfinite(img).all()),
            "kernels": {k: {"ms_per_launch": v["ms"] / max(1, v["launches"]), "launches": v["launches"],
                            "algorithmic_bytes_per_pixel_sample": work["bytes_per_pixel_sample"].get(k)}
                        for k, v in kernels.items()},
            "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "ms_per_launch": dom_ms,
                         "node_counts_from": "cpu oracle tree (64-byte binary nodes)" if ref_tree else "device (four-wide node visits priced at 64 B)",
                         # the same kernel with the traversal priced by the device tree's own visits, and without traversal
                         "frac_device_counts": dom["frac_device_counts"], "frac_compulsory": dom["frac_compulsory"],
                         "frac_traffic": dom.get("frac_traffic"),
                         # the kernels by total HIP-event time (under two batch lanes a launch's duration includes waiting
                         # for room beside the other lane's kernels, so the order of close candidates differs between runs)
                         "dominant_by": "largest kernel total of a one-lane pass" if alone else "largest kernel total of the timed region",
                         "candidates": lines[:4],
                         "pipeline": pipeline},
            "batch": {"pixel_samples": batch_ps, "lanes": lanes, "wavefront_depth": wf_depth},
            "create_seconds": st["build_seconds"],   # ubdpt_create: flattening, upload, tree build, batch buffers
        }
        if cb is not None:
            out["cpu_baseline"] = cb
            # relMSE of the GPU image against the CPU oracle at equal spp (same passes, same seed), box scenes only
            if total_passes <= 32 and args.config not in MESH_CONFIGS:
                ref = orc.render_passes(0, total_passes, args.seed, num_threads=cb["cores"])
                g, r = img / total_passes, ref / total_passes
                out["relmse_vs_cpu_equal_spp"] = float(np.mean((g - r) ** 2 / (r ** 2 + 1e-2)))
    ctx.close()   # (a second close does nothing)
    if rank == 0:
        if ngpu == 1 and not args.no_other_configs:
            out["other_configs"] = {}
            for c in ("C2", "C3", "C4", "C5"):
                if c != args.config:
                    try:
                        out["other_configs"][c] = short_run(c, device, args.seed)
                    except Exception as e:  # a configuration that does not fit is reported, not hidden
                        out["other_configs"][c] = {"error": str(e)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
