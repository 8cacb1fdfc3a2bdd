#!/usr/bin/env python3
"""Headline benchmark: BDPT throughput (Mpaths/s) of the HIP backend on the C2 workload.

  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Cornell box with a dielectric and a perfect-mirror sphere,
512x512. One "step" = one ubdpt_render_passes call of --passes-per-step progressive passes
(default 16); the default 16 steps therefore render the configuration's 256 spp on one GPU.
Inputs (scene, tree, sum buffer) are resident in HBM before the timed region.

N > 1 (launched by torch.distributed.run, one rank per GPU): rank r renders passes r, r+N, r+2N, ...
(pass_stride = N, no data-path collective while rendering); after the last step the per-GPU sum
buffers are combined by one RCCL reduce. Work per GPU is fixed => "scaling": "weak".

The JSON line also carries
  roofline     — algorithmic HBM bytes of the dominant kernel / its HIP-event time / 8 TB/s
  cpu_baseline — the CPU oracle (restatement of the reference renderer) timed on this host on a
                 bounded number of passes of the same workload.
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


def algorithmic_bytes(k):
    """Algorithmic bytes of one kernel's counted work: closest-hit ray 32 B in + 32 B out, shadow ray
    32 B in + 4 B out, 64 B per tree node visited, 48 B per primitive tested, 96 B per path vertex
    written (subpath kernels) or read (connection kernel), 24 B per framebuffer read-modify-write."""
    return (64.0 * k["closest_hit_rays"] + 36.0 * k["shadow_rays"] + 64.0 * k["nodes_visited"]
            + 48.0 * k["prims_tested"] + 96.0 * k["vertices"] + 24.0 * k["fb_updates"])


def measure_work(scene, params, device, passes, seed, accel, pipeline=0):
    """Counts the work of `passes` passes with the counting build of the kernels (deterministic)."""
    opt = capi.Options()
    opt.collect_stats = 1
    opt.accel = accel
    opt.pipeline = pipeline
    with capi.Context(scene, params, opt, device) as ctx:
        ctx.render_passes(0, passes, seed)
        s = ctx.stats()
    ps = float(s["pixel_samples"])
    per_kernel = {}
    ev = s["eye_vertices"]; lv = s["light_vertices"]
    # vertex records: the generation kernel writes the lens and the emitter vertex of every
    # pixel-sample, the bounce / tail / per-path kernels the rest (split by their share of closest-hit
    # rays); the connection stage reads every vertex once
    # k_wf_shade pushes the vertices of the hits that k_wf_trace found, so its share is trace's rays
    by_name = {k["name"]: k for k in s["kernels"]}
    writer_rays = {}
    for name in ("k_wf_bounce", "k_wf_tail", "k_eye", "k_light"):
        if name in by_name:
            writer_rays[name] = float(by_name[name]["closest_hit_rays"])
    if "k_wf_shade" in by_name and "k_wf_trace" in by_name:
        writer_rays["k_wf_shade"] = float(by_name["k_wf_trace"]["closest_hit_rays"])
    wrays = sum(writer_rays.values()) or 1.0
    has_gen = "k_wf_gen" in by_name
    rest = (ev + lv - 2 * ps) if has_gen else (ev + lv)
    # the connection stage reads every vertex once; with the split pair kernels that read is charged
    # to the evaluation kernel (the visibility kernel is charged its shadow rays and their traversal)
    readers = ("k_connect", "k_pair_process", "k_pair_eval")
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
        per_kernel[k["name"]] = algorithmic_bytes(kk) / ps
    return {
        "bytes_per_pixel_sample": per_kernel,
        "rays_per_pixel_sample": (s["closest_hit_rays"] + s["shadow_rays"]) / ps,
        "closest_per_pixel_sample": s["closest_hit_rays"] / ps,
        "shadow_per_pixel_sample": s["shadow_rays"] / ps,
        "nodes_per_ray": s["bvh_nodes_visited"] / max(1.0, s["closest_hit_rays"] + s["shadow_rays"]),
        "prims_per_ray": s["prims_tested"] / max(1.0, s["closest_hit_rays"] + s["shadow_rays"]),
        "eye_vertices": ev / ps, "light_vertices": lv / ps, "splats": s["splats"] / ps,
    }


def cpu_baseline(scene, params, seed, budget_s=12.0):
    """Times the CPU oracle on the same workload: one calibration pass, then as many passes as fit
    the budget (at most 64). All host cores the box grants this job (OMP_NUM_THREADS, default 16)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    cores = max(1, min(cores, os.cpu_count() or 1))
    orc = oracle_py.Oracle(scene, params, 32)
    npix = int(params.img_res_x * params.img_res_y)
    t = time.perf_counter()
    orc.render_passes(0, 1, seed, num_threads=cores)
    t1 = time.perf_counter() - t
    n = int(max(1, min(64, budget_s / max(t1, 1e-3))))
    t = time.perf_counter()
    img, cn = orc.render_passes(1, n, seed, num_threads=cores, counters=True)
    dt = time.perf_counter() - t
    return {
        "value": npix * n / dt / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
        "sample": f"{n} passes of the same 512x512 C2 workload ({npix * n} pixel-samples, {dt:.1f} s), "
                  f"linear-scan fp64 restatement of the reference, {cores} threads",
        "mrays_per_s": (cn["closest_hit_rays"] + cn["shadow_rays"]) / dt / 1e6,
    }, orc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--passes-per-step", type=int, default=16)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--accel", type=int, default=0)
    ap.add_argument("--pipeline", type=int, default=0)
    ap.add_argument("--passes-in-flight", type=int, default=0)
    ap.add_argument("--dense-threshold", type=int, default=0,
                    help="lanes of a wave that must own an (s,t) pair for it to be connected in place (0 = default)")
    ap.add_argument("--wavefront-depth", type=int, default=0, help="bounces run as wavefront launches (0 = default)")
    ap.add_argument("--streams", type=int, default=0, help="batches in flight on separate streams (0 = default)")
    ap.add_argument("--batch-pixel-samples", type=int, default=0, help="sub-pass batches of this many pixels (0 = whole passes)")
    ap.add_argument("--shadow-ray-budget", type=lambda x: int(x, 0), default=0,
                    help="large scenes: traversal steps before a shadow ray is deferred to a second launch (0 = default, 0xffffffff = off)")
    ap.add_argument("--closest-ray-budget", type=lambda x: int(x, 0), default=0,
                    help="large scenes: the same for the closest-hit rays of the subpaths (0 = default, 0xffffffff = off)")
    ap.add_argument("--seed", type=lambda x: int(x, 0), default=0x5EED)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    pps = args.passes_per_step

    work = None
    if rank == 0:
        work = measure_work(scene, params, device, 4, args.seed, args.accel, args.pipeline)

    opt = capi.Options()
    opt.accel = args.accel
    opt.pipeline = args.pipeline
    opt.passes_in_flight = args.passes_in_flight
    opt.dense_threshold = args.dense_threshold
    opt.wavefront_depth = args.wavefront_depth
    opt.streams = args.streams
    opt.batch_pixel_samples = args.batch_pixel_samples
    opt.shadow_ray_budget = args.shadow_ray_budget
    opt.closest_ray_budget = args.closest_ray_budget
    ctx = capi.Context(scene, params, opt, device)
    sum_dev = torch.zeros((H, W, 3), dtype=torch.float64, device=f"cuda:{device}")

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
    for s in range(args.steps):
        # step s covers passes [s * pps * ngpu, (s + 1) * pps * ngpu); this rank takes every ngpu-th of them
        first, count, stride = multigpu.partition_passes(s * pps * ngpu, pps * ngpu, rank, ngpu)
        ctx.enqueue_passes(first, count, args.seed, pass_stride=stride)
    ctx.read_sum_device(sum_dev.data_ptr())  # waits for the enqueued passes
    multigpu.reduce_sum(sum_dev, dist)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{device}" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        total_passes = args.steps * pps * ngpu
        pixel_samples = total_passes * npix
        value = pixel_samples / elapsed / 1e6
        st = ctx.stats()
        kernels = {k["name"]: k for k in st["kernels"]}
        dom = max((k for k in kernels.values() if work["bytes_per_pixel_sample"].get(k["name"])), key=lambda k: k["ms"])
        dom_ms = dom["ms"] / max(1, dom["launches"])
        # bytes of one launch = bytes per pixel-sample x pixel-samples of one batch / launches per batch
        n_batches = max(1, min(k["launches"] for k in kernels.values()))
        dom_bytes = work["bytes_per_pixel_sample"].get(dom["name"], 0.0) * (pixel_samples / ngpu) / max(1, dom["launches"])
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(dom["name"])
            except Exception:
                traffic = None
        img = sum_dev.cpu().numpy()
        out = {
            "metric": "bdpt_mpaths_per_s", "value": value, "unit": "Mpaths/s", "n_gpus": ngpu, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {WORKLOADS.get(args.config, 'synthetic scene')}, {W}x{H}",
                       "passes_per_step": pps, "spp_total": total_passes, "passes_per_gpu": args.steps * pps,
                       "accel": "lbvh" if args.accel == 0 else "linear", "pipeline": args.pipeline,
                       "passes_in_flight": args.passes_in_flight or "auto", "streams": args.streams or "default",
                       "wavefront_depth": args.wavefront_depth or "default", "partition": f"passes round-robin over {ngpu} GPU(s)"},
            "mrays_per_s": value * work["rays_per_pixel_sample"],
            "rays_per_pixel_sample": work["rays_per_pixel_sample"],
            "work_per_pixel_sample": {k: work[k] for k in ("closest_per_pixel_sample", "shadow_per_pixel_sample",
                                                           "nodes_per_ray", "prims_per_ray", "eye_vertices",
                                                           "light_vertices", "splats")},
            "mean_radiance": [float(x) for x in (img.reshape(-1, 3).mean(0) / total_passes)],
            "finite": bool(np.isfinite(img).all()),
            "kernels": {k: {"ms_per_launch": v["ms"] / max(1, v["launches"]), "launches": v["launches"],
                            "algorithmic_bytes_per_pixel_sample": work["bytes_per_pixel_sample"].get(k)}
                        for k, v in kernels.items()},
            "roofline": {"bound": "hbm", "kernel": dom["name"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "ms_per_launch": dom_ms},
        }
        if ngpu == 1 and not args.no_cpu_baseline:
            cb, orc = cpu_baseline(scene, params, args.seed)
            out["cpu_baseline"] = cb
            # relMSE of the GPU image against the CPU oracle at equal spp (same passes, same seed)
            if total_passes <= 32:
                ref = orc.render_passes(0, total_passes, args.seed, num_threads=cb["cores"])
                g, r = img / total_passes, ref / total_passes
                out["relmse_vs_cpu_equal_spp"] = float(np.mean((g - r) ** 2 / (r ** 2 + 1e-2)))
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
