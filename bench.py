"""bench.py — BASELINE.json's metric on its own config.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): assets house.toml, 1920x1080, 256 spp, 8 bounces, synthetic 2048x1024
environment (the reference's HDRIs are missing blobs) — BASELINE.json configs[3], the
configuration `metric` is quoted on.  A step = one complete frame (all 256 samples of every
pixel) of that workload through the C-ABI, inputs resident in HBM.  With N GPUs the SAME frame is
partitioned by interleaved 16x16 tiles (one process per GPU) and one RCCL reduce(sum) brings the
accumulators to rank 0 inside the timed region ("scaling": "strong").

metric = Mrays/s = (extension rays + issued shadow rays, counted on the device) / wall time;
ms_per_step = ms/frame.  roofline + cpu_baseline as the round contract asks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
COUNTS_FILE = os.path.join(ROOT, "profiles", "algo_counts_house_1080p_8b.json")
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "hbm_traffic_house_1080p_8b.json")  # rocprofv3 PMC, tools/profile.sh


def algorithmic_bytes(st):
    """SURVEY.md §8(d) per-unit figures applied to instrumented counts (unpruned, any-hit shadow).
    Returns bytes for everything in `st` except the once-per-pixel accumulator write."""
    rays = st["ext_rays"] + st["shadow_rays"]
    b = 32 * st["nodes_visited"] + 8 * st["prim_refs"]
    b += 20 * (st["sphere_tests"] + st["fallback_sphere_tests"]) + 64 * (st["plane_tests"] + st["fallback_plane_tests"])
    b += 64 * st["tri_tests"] + 36 * st["closest_tri"] + 32 * rays
    b += (48 + 2 * 16 + 64) * st["nee_events"] + (64 + 16) * st["escapes"]
    return b


def cpu_leg(scene, env, width, height, bounces, log):
    """cpu_baseline leg (rank 0, N = 1 only): the oracle port on the host cores, bounded sample."""
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    osc, oenv, cam = util.oracle_scene(scene), util.oracle_env(env), scene.camera_uniform().view(oracle.CAMERA)
    cores = os.cpu_count() or 1
    # counts for the roofline: the traversal the kernel executes — every node the reference visits for
    # extension rays, any-hit exit for shadow rays (pruning is not exactly result-preserving)
    t = time.time()
    _, counts = oracle.render(osc, oenv, cam, width, height, 0, 1, bounces, flags=oracle.FLAG_ANYHIT_SHADOW,
                              n_threads=cores, fast=True)
    t_counts = time.time() - t
    # timed baseline: the reference's own traversal, sized for roughly 15 s (calibrated on 2 spp)
    t = time.time()
    oracle.render(osc, oenv, cam, width, height, 0, 2, bounces, flags=0, n_threads=cores, fast=True)
    t_cal = (time.time() - t) / 2
    spp = int(max(2, min(256, round(20.0 / max(t_cal, 0.01)))))
    t = time.time()
    _, st = oracle.render(osc, oenv, cam, width, height, 0, spp, bounces, flags=0, n_threads=cores, fast=True)
    dt = time.time() - t
    rays = st["ext_rays"] + st["shadow_rays"]
    log("cpu_baseline: %d spp in %.2f s on %d threads = %.2f Mrays/s (counting pass 1 spp: %.2f s)" % (spp, dt, cores, rays / dt / 1e6, t_counts))
    per_path = algorithmic_bytes(counts) / counts["paths"]
    base = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "house.toml %dx%d, samples 0..%d of every pixel, %d bounces, reference traversal, liboracle_fast.so (-O3, OpenMP dynamic 16x16 tiles)"
                      % (width, height, spp - 1, bounces),
            "ms_per_frame_extrapolated": dt / spp * 256 * 1e3, "cpu_model": cpu_model(), "seconds": dt}
    return base, per_path, counts


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--scene", default="house")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; the real thing) or gloo (rehearsal of N > 1 on one GPU)")
    ap.add_argument("--write-counts", action="store_true", help="store the per-path algorithmic bytes under profiles/")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the integrator has no CPU path")
    device_index = local_rank % torch.cuda.device_count()  # == local_rank on a real N-GPU node
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import rsoderh_raytracing_amd as R
    from rsoderh_raytracing_amd import partition
    scene_file = os.path.join(ROOT, "tests", "golden", "assets", "scenes", args.scene + ".toml")
    scene = R.Scene.load_toml(scene_file)
    env = R.Environment.synthetic(2048, 1024)
    W, H, spp = args.width, args.height, args.spp

    state = R.State.new(scene, env, W, H, device=device_index)
    state.max_bounces = args.bounces
    state.set_partition(rank, world, partition.TILE_W, partition.TILE_H)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    state.bind_accumulator(acc.data_ptr(), W, H)
    stream = torch.cuda.Stream()
    log(state.describe())

    def step():
        acc.zero_()
        state.render_range(0, spp, stream=stream.cuda_stream)
        if world > 1:
            if args.backend == "nccl":
                partition.reduce_accumulators(acc)  # one RCCL reduce(sum) over xGMI
            else:  # rehearsal: gloo reduces on the host
                host = acc.cpu()
                partition.reduce_accumulators(host)
                if rank == 0:
                    acc.copy_(host)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        fence()
        state.stats()  # drop warm-up counters
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        fence()
        elapsed = time.perf_counter() - t0

    s = state.stats()
    cdev = "cuda" if args.backend == "nccl" else "cpu"
    tot = torch.tensor([float(s["ext_rays"] + s["shadow_rays"]), float(s["paths"])], dtype=torch.float64, device=cdev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    rays_total, paths_total = float(tot[0].item()), float(tot[1].item())

    result = None
    if rank == 0:
        frame = acc.cpu().numpy()
        assert np.isfinite(frame).all() and np.all(frame[..., 3] == 1.0), "frame incomplete"
        value = rays_total / elapsed / 1e6
        launches_trace = args.steps * max(1, s["launches"] // (2 * args.steps))
        trace_ms_per_launch = s["trace_kernel_ms"] / launches_trace
        cpu_base, per_path = None, None
        std_cfg = (args.scene, W, H, args.bounces) == ("house", 1920, 1080, 8)
        if world == 1 and not args.no_cpu_baseline:
            cpu_base, per_path, counts = cpu_leg(scene, env, W, H, args.bounces, log)
            if args.write_counts and std_cfg:
                os.makedirs(os.path.dirname(COUNTS_FILE), exist_ok=True)
                with open(COUNTS_FILE, "w") as f:
                    json.dump({"per_path_algorithmic_bytes": per_path, "counts_1spp": counts,
                               "source": "oracle, unpruned extension rays + any-hit shadow rays, sample 0 of every pixel"}, f, indent=1)
        elif std_cfg and os.path.exists(COUNTS_FILE):
            with open(COUNTS_FILE) as f:
                per_path = json.load(f)["per_path_algorithmic_bytes"]
        roofline = None
        if per_path is not None:
            paths_per_launch = s["paths"] / launches_trace  # this rank's kernel
            owned_pixels = int(partition.owned_mask(W, H, rank, world).sum())
            algo = per_path * paths_per_launch + 16.0 * owned_pixels
            achieved = algo / (trace_ms_per_launch * 1e-3) / 1e9
            traffic = issue = None  # PMC counters cannot be read from inside this process: committed rocprofv3 measurement
            if std_cfg and world == 1 and spp == 256 and os.path.exists(TRAFFIC_FILE):
                with open(TRAFFIC_FILE) as f:
                    tj = json.load(f)
                traffic = (tj.get("fetch_bytes_per_launch") or 0) + (tj.get("write_bytes_per_launch") or 0)
                issue = tj.get("issue")  # what actually bounds the kernel: VALU issue (same rocprofv3 run)
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "kernel": "rt_render_pool_kernel", "launch_ms": trace_ms_per_launch,
                        "algorithmic_bytes_per_launch": algo, "algorithmic_bytes_per_path": per_path,
                        "sample_buffer_bytes_per_launch": 24.0 * paths_per_launch, "valu_issue": issue,
                        "note": "working set (10 KB scene image in LDS + 64 MiB environment, MALL-resident) is cache-resident by "
                                "construction, so the fraction of the HBM peak exceeds 1; the kernel is bound by FP32 VALU issue "
                                "and lane divergence (valu_issue, DESIGN.md section 5)"}
        result = {"metric": "Mrays/s, house.toml 1920x1080 256spp 8-bounce", "value": value, "unit": "Mrays/s",
                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                  "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                  "config": {"workload": "%s.toml %dx%d %d spp %d bounces, synthetic 2048x1024 HDRI env" % (args.scene, W, H, spp, args.bounces),
                             "parallelism": "tiles16x16-interleaved x%d + rccl reduce" % world if world > 1 else "single GPU",
                             "rays_per_frame": rays_total / args.steps, "paths_per_frame": paths_total / args.steps},
                  "ms_per_frame": elapsed / args.steps * 1e3,
                  "kernel_ms_per_frame": {"trace": s["trace_kernel_ms"] / args.steps, "resolve": s["resolve_kernel_ms"] / args.steps},
                  "roofline": roofline, "cpu_baseline": cpu_base}
    state.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if result is not None:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
