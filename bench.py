"""bench.py — BASELINE.json's metric on its own config.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): assets house.toml, 1920x1080, 256 spp, 8 bounces, synthetic 2048x1024
environment (the reference's HDRIs are missing blobs) — BASELINE.json configs[3], the
configuration `metric` is quoted on.  A step = one complete frame (all 256 samples of every
pixel) of that workload through the C-ABI, inputs resident in HBM.  With N GPUs the SAME frame is
partitioned by interleaved 16x16 tiles (one process per GPU) and ONE RCCL exchange — every rank's own tiles, 1/N of the
frame, gathered by librsrt itself (rsrt_comm_reduce, include/rsrt.h) — brings the frame to rank 0 inside the
timed region ("scaling": "strong").  torch.distributed (gloo) is only the control plane: barrier,
hand-over of the RCCL id, sums of the counters.

metric = Mrays/s = (extension rays + issued shadow rays, counted on the device) / wall time;
ms_per_step = ms/frame.

N > 1 proves itself and cannot hang (round 4): after the timed region rank 0 renders the WHOLE frame alone into a second buffer BEFORE the timed
region and every candidate exchange — librsrt's RCCL gather of compact tile buffers, its dense ncclReduce, torch's nccl reduce, gloo — must reproduce
it bit for bit before it is timed (the first that does is used, all ranks switching together: `multi_gpu.pre_flight`); the JSON line says whether
the last timed frame equals it too (`frame_equals_1gpu`), which exchange really ran, and every rank's trace / exchange milliseconds (`multi_gpu`); the collective bring-up (rsrt_comm_init) and the first exchange run under a watchdog
thread that, should they not return within RSRT_BENCH_WATCHDOG_S seconds (default 120), prints rank / step / an NCCL_DEBUG hint to stderr
and ends the process with os._exit(3) — a fresh non-zero exit, never a re-exec.  At N = 1 the line also carries `extra_configs`: the other
BASELINE.json configs (default, cube, suzanne, the 15 k-triangle grid, sixteen back-to-back single-sample calls), 3 frames each AFTER the
timed region, with the oracle-counted roofline fraction of each.

roofline: the kernel's scene is LDS-resident and its environment MALL-resident, so HBM is not the
roof it is under; FP32 VALU issue is.  `frac` = ALGORITHMIC lane-operations per second (the f32 / u32 operations the
reference's integrator executes per path, counted by the oracle's counting build, x paths / launch time) / 78.64 T/s;
`utilisation` = the kernel's own retired VALU lane-instructions per SIMD-cycle / 32 (a SIMD retires at most one
wave64 VALU instruction per 2 cycles = 32 lanes per cycle), from rocprofv3 PMC counters collected
LIVE by this run: before the GPU is touched, rank 0 (N = 1) starts `rocprofv3 --kernel-trace --pmc ... -- python3
bench.py --pmc-child` four times (separate passes: SQ + GRBM, FETCH_SIZE + LDS, WRITE_SIZE + TCC hit / miss, read
requests by size), each rendering one frame of the same workload with the same library.  If rocprofv3 cannot run, the committed counters of
profiles/pmc_house_1080p_8b.json are used — but only when they were taken on the same kernel sources
(rsrt_build_id); otherwise the object says so and carries no fraction.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_SIMD = 256 * 4        # 256 CUs x 4 SIMD32
LANES_PER_SIMD_CYCLE = 32.0  # wave64 VALU instruction = 2 cycles on a SIMD32 (guide: v_fma_f32 2 cyc)
MAX_CLOCK_HZ = 2.4e9
LDS_ARRAY_CYCLES_PER_CU_CYCLE = 1.0
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_house_1080p_8b.json")
# NO TA_* / TD_* counters in any pass: a rocprofv3 --pmc run with them hung on this pool and cost a 15-minute GPU call (round 2).  AVOIDED, NOT
# ROOT-CAUSED: the killed call merged no file back (gpurun_out/ holds nothing of it — checked again in round 4), so which dispatch was in flight,
# and whether the render kernel itself had finished, is not on record; all that is known is that the same kernel, frame and rocprofv3 passes without
# those counters have completed in every one of the ~60 PMC runs since.  tools/pmc_scene.sh passes its extra lists through here and refuses them too.
PMC_PASSES = [  # pass 0 is what `utilisation` needs; the others are reported when they succeed
    ["SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
     "SQ_BUSY_CYCLES", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE"],
    ["FETCH_SIZE", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
     "SQ_WAVES"],
    ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],  # the read requests by size
]


if os.environ.get("RSRT_PMC_EXTRA"):  # diagnosis: more passes, e.g. "TCP_TOTAL_CACHE_ACCESSES_sum,TCP_GATE_EN1_sum" (NOT TA_* / TD_*: they hang rocprofv3 here)
    PMC_PASSES = PMC_PASSES + [p.split(",") for p in os.environ["RSRT_PMC_EXTRA"].split(";") if p and "TA_" not in p and "TD_" not in p]


def algorithmic_bytes(st):
    """SURVEY.md §8(d) per-unit figures applied to instrumented counts (unpruned, any-hit shadow).
    Returns bytes for everything in `st` except the once-per-pixel accumulator write."""
    rays = st["ext_rays"] + st["shadow_rays"]
    b = 32 * st["nodes_visited"] + 8 * st["prim_refs"]
    b += 20 * (st["sphere_tests"] + st["fallback_sphere_tests"]) + 64 * (st["plane_tests"] + st["fallback_plane_tests"])
    b += 64 * st["tri_tests"] + 36 * st["closest_tri"] + 32 * rays
    b += (48 + 2 * 16 + 64) * st["nee_events"] + (64 + 16) * st["escapes"]
    return b


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def effective_cores():
    """The cores this process may really use: logical CPUs, capped by the affinity mask and by the cgroup's CPU quota
    (the GPU box shows 256 logical CPUs of an EPYC 9575F pair to a container that is given 16 of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = parts[0], float(parts[1])
            else:
                quota = parts[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                    period = float(f2.read().split()[0])
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(-(-float(quota) // period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_leg(scene, env, width, height, bounces, log):
    """cpu_baseline leg (rank 0, N = 1 only): the oracle port on the host cores, bounded sample.  BASELINE.md §2:
    config 1 (default.toml 256x256, 4 spp, 3 bounces) timed in full; the bench workload on >= 4 spp, scaled."""
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    import rsoderh_raytracing_amd as R
    osc, oenv, cam = util.oracle_scene(scene), util.oracle_env(env), scene.camera_uniform().view(oracle.CAMERA)
    cores = effective_cores()
    # counts for the algorithmic-byte model: the traversal the kernel executes — every node the reference visits for
    # extension rays, any-hit exit for shadow rays (pruning is not exactly result-preserving)
    t = time.time()
    # ... through the counting build (liboracle_ops.so: same bits + the f32 / u32 operations the integrator executes)
    _, counts = oracle.render(osc, oenv, cam, width, height, 0, 1, bounces, flags=oracle.FLAG_ANYHIT_SHADOW,
                              n_threads=cores, fast="ops")
    counts["leaf_boxes"] = int((np.asarray(scene.bvh_nodes["primitives_len"]) > 0).sum())
    t_counts = time.time() - t
    # timed baseline: the reference's own traversal, sized for roughly 15-20 s (calibrated on 2 spp)
    t = time.time()
    oracle.render(osc, oenv, cam, width, height, 0, 2, bounces, flags=0, n_threads=cores, fast=True)
    t_cal = (time.time() - t) / 2
    spp = int(max(4, min(256, round(20.0 / max(t_cal, 0.01)))))
    t = time.time()
    _, st = oracle.render(osc, oenv, cam, width, height, 0, spp, bounces, flags=0, n_threads=cores, fast=True)
    dt = time.time() - t
    rays = st["ext_rays"] + st["shadow_rays"]
    # config 1, complete, best of 3 (it takes milliseconds)
    d_sc = R.Scene.load_toml(os.path.join(ROOT, "tests", "golden", "assets", "scenes", "default.toml"))
    d_osc, d_cam = util.oracle_scene(d_sc), d_sc.camera_uniform().view(oracle.CAMERA)
    best, c1 = None, None
    for _ in range(3):
        t = time.time()
        _, c1 = oracle.render(d_osc, oenv, d_cam, 256, 256, 0, 4, 3, flags=0, n_threads=cores, fast=True)
        best = min(best, time.time() - t) if best else time.time() - t
    c1_rays = c1["ext_rays"] + c1["shadow_rays"]
    # one core alone, for the per-core figure (threads do not scale linearly across sockets / SMT)
    t = time.time()
    _, s1 = oracle.render(osc, oenv, cam, width // 4, height // 4, 0, 1, bounces, flags=0, n_threads=1, fast=True)
    dt1 = time.time() - t
    log("cpu_baseline: %d spp in %.2f s on %d threads = %.2f Mrays/s (%.3f per thread; one thread alone %.3f); config 1 in %.1f ms; counting pass %.2f s"
        % (spp, dt, cores, rays / dt / 1e6, rays / dt / 1e6 / cores, (s1["ext_rays"] + s1["shadow_rays"]) / dt1 / 1e6, best * 1e3, t_counts))
    per_path = algorithmic_bytes(counts) / counts["paths"]
    base = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "house.toml %dx%d, samples 0..%d of every pixel, %d bounces, reference traversal, liboracle_fast.so (-O3, OpenMP dynamic 16x16 tiles)"
                      % (width, height, spp - 1, bounces),
            "per_core_mrays_s": rays / dt / 1e6 / cores,
            "single_thread_mrays_s": (s1["ext_rays"] + s1["shadow_rays"]) / dt1 / 1e6,
            "ms_per_frame_extrapolated": dt / spp * 256 * 1e3, "extrapolation_factor": 256.0 / spp,
            "config1_default_256x256_4spp_3b": {"ms": best * 1e3, "mrays_s": c1_rays / best / 1e6, "rays": c1_rays, "timed": "in full, best of 3"},
            "cpu_model": cpu_model(), "logical_cpus": os.cpu_count(), "seconds": dt,
            "cores_note": "threads = logical CPUs capped by the affinity mask and the container's CPU quota"}
    return base, per_path, counts


def scene_path(name):
    """--scene: a name under tests/golden/assets/scenes/ or the path of a .toml (tools/make_big_scene.py)."""
    return name if name.endswith(".toml") else os.path.join(ROOT, "tests", "golden", "assets", "scenes", name + ".toml")


# ---------------------------------------------------------------------------------------------- PMC (rocprofv3)
def pmc_child(args):
    """Runs under rocprofv3: ONE frame of the workload through the C-ABI, nothing else (no torch)."""
    import rsoderh_raytracing_amd as R
    from rsoderh_raytracing_amd import state as S
    scene = R.Scene.load_toml(scene_path(args.scene))
    env = R.Environment.synthetic(2048, 1024)
    st = R.State.new(scene, env, args.width, args.height, device=0)
    st.max_bounces = args.bounces
    st.render_range(0, args.spp)
    st.synchronize()
    s = st.stats()
    st.close()
    print(json.dumps({"build_id": S.build_id(), "trace_kernel_ms": s["trace_kernel_ms"], "paths": s["paths"]}), flush=True)


def collect_pmc(args, log):
    """Three rocprofv3 --pmc passes of the child.  Returns {"counters": {...}, "kernel": name, "build_id": ..} or None."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        log("pmc: rocprofv3 not found")
        return None
    out = {"counters": {}, "resolve": {}, "kernel": None, "build_id": None, "launch_ms_under_pmc": None}
    tmp = tempfile.mkdtemp(prefix="rsrt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for i, counters in enumerate(PMC_PASSES):
            try:
                d = os.path.join(tmp, "pass%d" % i)
                cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                                                                     "--pmc-child", "--scene", args.scene, "--width", str(args.width), "--height", str(args.height),
                                                                     "--spp", str(args.spp), "--bounces", str(args.bounces)]
                t = time.time()
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=90)
                if r.returncode != 0:
                    raise RuntimeError("rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:]))
                child = None
                for line in r.stdout.splitlines():
                    if line.startswith("{") and "build_id" in line:
                        child = json.loads(line)
                if child is None:
                    raise RuntimeError("child printed no result line")
                if out["build_id"] not in (None, child["build_id"]):
                    raise RuntimeError("library changed between passes")
                out["build_id"] = child["build_id"]
                rows = 0
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    with open(f) as fh:
                        for row in csv.DictReader(fh):
                            name = row["Kernel_Name"]
                            if "rt_render_pool_kernel" in name or "rt_render_kernel" in name:
                                out["kernel"] = name
                                out["counters"][row["Counter_Name"]] = out["counters"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                                if i == 0:
                                    out["launch_ms_under_pmc"] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
                                rows += 1
                            elif "rt_resolve_kernel" in name:
                                out["resolve"][row["Counter_Name"]] = out["resolve"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                if rows == 0:
                    raise RuntimeError("no counter rows for the render kernel")
                log("pmc pass %d (%s ...): %.1f s" % (i, counters[0], time.time() - t))
            except (OSError, subprocess.SubprocessError, ValueError, KeyError, RuntimeError) as e:
                log("pmc pass %d (%s ...) failed: %s" % (i, counters[0], e))
                if i == 0:
                    return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["source"] = "live: rocprofv3 --kernel-trace --pmc, %d separate passes of one frame each, started by this bench run" % len(PMC_PASSES)
    return out


def roofline_object(pmc, build_id, launch_ms, paths_per_launch, per_path, owned_pixels, spp, counts=None):
    """The roofline of the dominant kernel.  pmc: collect_pmc()'s result (or the committed one) or None; counts: the
    counting oracle's totals over sample 0 of every pixel (f32_ops, int_ops, paths, rays, nodes ...) or None.

    frac        = ALGORITHMIC lane-operations per second / peak: (f32 + u32 operations per path, counted by the oracle's
                  counting build on the reference's own walk) x paths per launch / launch time / (1,024 SIMDs x 32 lanes x
                  2.4 GHz = 78.64 T/s).  What the reference's integrator needs, whatever the kernel executes for it.
    utilisation = the kernel's own retired VALU lane-instructions per SIMD-cycle / 32 (counters): how busy the lanes are,
                  scheduler, extra box tests and multi-instruction div / sqrt included.
    overhead    = 1 - frac / utilisation: the share of the retired lane-instructions that is not algorithmic work."""
    algo = None
    if per_path is not None:
        algo_bytes = per_path * paths_per_launch + 16.0 * owned_pixels
        algo = {"bytes_per_launch": algo_bytes, "bytes_per_path": per_path, "GB_per_s": algo_bytes / (launch_ms * 1e-3) / 1e9,
                "frac_of_hbm_peak": algo_bytes / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "counts": "SURVEY §8(d) per-unit bytes x counts of the instrumented oracle on sample 0 of every pixel, UNPRUNED extension rays "
                          "(t-pruning changes 2 of 5.3e8 paths, DESIGN.md §2: the kernel tests what the reference tests) + any-hit shadow rays",
                "note": "these bytes are served by LDS (scene) and L2 / Infinity Cache (environment): a fraction of the HBM peak above 1 "
                        "means HBM is not the roof"}
    peak_tops = N_SIMD * LANES_PER_SIMD_CYCLE * MAX_CLOCK_HZ / 1e12
    ro = {"bound": "valu", "achieved": None, "peak": peak_tops, "unit": "T lane-operations/s", "frac": None, "utilisation": None, "overhead": None,
          "traffic": None, "kernel": None, "build_id": build_id, "launch_ms": launch_ms, "sample_buffer_bytes_per_launch": 12.0 * paths_per_launch,
          "algorithmic": algo}
    if counts is not None and counts.get("f32_ops"):
        ops_per_path = (counts["f32_ops"] + counts["int_ops"]) / counts["paths"]
        rays = counts["ext_rays"] + counts["shadow_rays"]
        achieved = ops_per_path * paths_per_launch / (launch_ms * 1e-3) / 1e12
        ro.update({"achieved": achieved, "frac": achieved / peak_tops,
                   "work": {"f32_ops_per_path": counts["f32_ops"] / counts["paths"], "u32_ops_per_path": counts["int_ops"] / counts["paths"],
                            "ops_per_launch": ops_per_path * paths_per_launch, "rays_per_path": rays / counts["paths"],
                            "reference_walk_nodes_per_ray": counts["nodes_visited"] / rays, "primitive_tests_per_ray": counts["prim_refs"] / rays,
                            "flat_loop_leaf_boxes_per_ray": counts.get("leaf_boxes"),
                            "counted_by": "oracle/liboracle_ops.so (-DORC_COUNT_OPS) on sample 0 of every pixel: one per f32 add / sub / mul / div / fma / sqrt / "
                                          "floor / min / max / comparison, detmath by polynomial path, u32 arithmetic and conversions apart; the reference's "
                                          "own walk (every node it visits) for extension rays, first-hit exit for shadow rays",
                            "note": "the flat loop tests every leaf box for every ray (flat_loop_leaf_boxes_per_ray) where the reference's walk visits "
                                    "reference_walk_nodes_per_ray nodes: more box tests, bought for full lanes; that difference, the scheduler, and div / sqrt / "
                                    "transcendentals costing several instructions each are what `overhead` holds"}})
    ledger_file = os.path.join(ROOT, "profiles", "r04_house_ledger.json")
    if counts is not None and counts.get("f32_ops") and os.path.exists(ledger_file):  # where the retired lane-instructions go, by cause (tools/ledger.py)
        with open(ledger_file) as f:
            lg = json.load(f)
        ro["overhead_ledger"] = {"lane_instructions_per_ray": lg["ledger_lane_instructions_per_ray"], "share": lg["ledger_share"],
                                 "counter_total_lane_instructions_per_ray": lg["counter_total_lane_instructions_per_ray"], "ablations": lg["ablations"],
                                 "trace_loops": lg["trace_loops"], "taken_on_build": lg["bench_build_id"], "workload": lg["workload"], "how": lg["how"],
                                 "reading": "no cause beyond the algorithmic operations holds much more than a tenth of what the kernel retires: IEEE forms 10 %, the flat loop's "
                                            "extra boxes 8 %, the scheduler 10 % (VALU; 12 % of the wave time), 16 % not attributed — the flat kernel is declared done at frac 0.24"}
    shares_file = os.path.join(ROOT, "profiles", "r04_house_stage_shares.json")
    if not os.path.exists(shares_file):
        shares_file = os.path.join(ROOT, "profiles", "r03_house_stage_shares.json")
    if counts is not None and counts.get("f32_ops") and os.path.exists(shares_file):  # what the non-algorithmic share consists of (instrumented build)
        with open(shares_file) as f:
            sh = json.load(f)
        ro["overhead_breakdown"] = {"wave_time_shares": sh["wave_time_shares"], "taken_on_build": sh["build_id"], "workload": sh["workload"], "source": sh["source"],
                                    "note": "census = the scheduler (stage census + compaction); GEN includes the fused first trace of the camera rays; the shares are "
                                            "of wave time, the algorithmic operations sit inside TRACE / SHADE / MISS / GEN"}
    if pmc is None:
        ro["note"] = "no PMC counters: rocprofv3 could not run here and no committed profile matches this library's build id (frac needs none; utilisation does)"
        return ro
    c = pmc["counters"]
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs (guide: DVFS give-back)
    insts = c["SQ_INSTS_VALU"]
    lanes = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]  # lanes active per VALU instruction (both in quad-cycles)
    ipc = insts / (cycles * N_SIMD)
    util = ipc * lanes / LANES_PER_SIMD_CYCLE
    ro.update({"utilisation": util, "overhead": (1.0 - ro["frac"] / util) if ro["frac"] is not None else None,
               "utilisation_unit": "retired VALU lane-instructions per SIMD-cycle / 32", "kernel": pmc["kernel"],
               "counters": {k: c[k] for k in sorted(c)}, "counters_source": pmc["source"], "counters_build_id": pmc["build_id"],
               "launch_ms_under_pmc": pmc.get("launch_ms_under_pmc"),
               "valu": {"wave_instructions_per_launch": insts, "instructions_per_simd_cycle": ipc, "issue_peak_per_simd_cycle": 0.5,
                        "issue_frac": ipc / 0.5, "lanes_active_per_instruction": lanes, "lane_frac": lanes / 64.0,
                        "clock_ghz_under_pmc": cycles / (pmc["launch_ms_under_pmc"] * 1e-3) / 1e9 if pmc.get("launch_ms_under_pmc") else None,
                        # the same work against the wall clock of THIS run's launches and the 2.4 GHz peak clock
                        "wall": {"achieved_tera_lane_instructions_per_s": insts * lanes / (launch_ms * 1e-3) / 1e12,
                                 "peak": N_SIMD * LANES_PER_SIMD_CYCLE * MAX_CLOCK_HZ / 1e12,
                                 "frac": insts * lanes / (launch_ms * 1e-3) / (N_SIMD * LANES_PER_SIMD_CYCLE * MAX_CLOCK_HZ)},
                        "wait_any_frac_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                        "wait_inst_any_frac_of_wave_cycles": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                        "salu_per_valu": c["SQ_INSTS_SALU"] / insts}})
    if "WRITE_SIZE" in c and ("FETCH_SIZE" in c or "TCC_EA0_RDREQ_128B_sum" in c):
        # WRITE_SIZE (KiB) is exact on gfx950; FETCH_SIZE = RDREQ x 64 B tallies a 128-byte read request as 64 bytes (guide,
        # HBM section; tools/fetch_calib.hip: coalesced 4 / 12 / 16-byte-per-lane streams read 1/2, a random 16-byte gather
        # is ONE 64-byte request and reads right).  Exact form: the requests by size, 128 x n128 + 64 x n64 + 32 x n32.
        # Cross-check: rt_resolve_kernel of the same pass streams a KNOWN 12 B x paths.  All at the L2's fabric side:
        # Infinity-Cache hits are included.
        write = c["WRITE_SIZE"] * 1024.0
        hbm = {"write_bytes_per_launch": write}
        if "TCC_EA0_RDREQ_128B_sum" in c:
            n128, n64, n32 = c["TCC_EA0_RDREQ_128B_sum"], c["TCC_EA0_RDREQ_64B_sum"], c["TCC_EA0_RDREQ_32B_sum"]
            other = c["TCC_EA0_RDREQ_sum"] - n128 - n64 - n32
            fetch = 128.0 * n128 + 64.0 * (n64 + max(other, 0.0)) + 32.0 * n32
            hbm.update({"fetch_bytes_per_launch": fetch, "fetch_from": "TCC_EA0_RDREQ by request size: 128 B x %.4g + 64 B x %.4g + 32 B x %.4g" % (n128, n64, n32)})
            if pmc["resolve"].get("TCC_EA0_RDREQ_128B_sum"):
                r = pmc["resolve"]
                hbm["resolve_kernel_check"] = {"known_bytes": 12.0 * paths_per_launch,
                                               "counted_bytes": 128.0 * r["TCC_EA0_RDREQ_128B_sum"] + 64.0 * r.get("TCC_EA0_RDREQ_64B_sum", 0.0) + 32.0 * r.get("TCC_EA0_RDREQ_32B_sum", 0.0)}
        else:
            fetch = c["FETCH_SIZE"] * 1024.0 * 2.0
            hbm.update({"fetch_bytes_per_launch": fetch, "fetch_from": "FETCH_SIZE x 2 (upper bound: every request taken as 128 bytes)"})
        if "FETCH_SIZE" in c:
            hbm["FETCH_SIZE_uncorrected_bytes"] = c["FETCH_SIZE"] * 1024.0
        ro["traffic"] = fetch + write
        hbm.update({"GB_per_s": (fetch + write) / (launch_ms * 1e-3) / 1e9, "frac_of_hbm_peak": (fetch + write) / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "compulsory_bytes_per_launch": 12.0 * paths_per_launch + 2 * 2048 * 1024 * 16 + 16.0 * owned_pixels,
                    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0 else None})
        ro["hbm"] = hbm
    if "SQ_LDS_IDX_ACTIVE" in c:
        ro["lds"] = {"array_busy_frac": c["SQ_LDS_IDX_ACTIVE"] / (cycles * 256 * LDS_ARRAY_CYCLES_PER_CU_CYCLE),
                     "bank_conflict_frac_of_busy": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], "wave_instructions_per_launch": c["SQ_INSTS_LDS"],
                     "note": "SQ_LDS_IDX_ACTIVE = LDS-array cycles summed over the 256 CUs; one array cycle moves up to 256 B"}
    ro["note"] = ("bound = FP32 VALU issue: the scene (10 KB) is read from LDS and the 64 MiB environment from L2 / Infinity Cache, so neither HBM "
                  "nor MFMA is the roof this kernel is under; frac = algorithmic lane-operations (counted by the oracle) per second / 78.64 T/s; "
                  "utilisation = (VALU wave-instructions per SIMD-cycle) x (lanes active per instruction) / 32, from live counters; frac <= utilisation <= 1")
    return ro


# ---------------------------------------------------------------------------------------------- N > 1: watchdog, proof
class watchdog:
    """`with watchdog("comm_init", rank, step): collective()` — a timer thread beside a call that may never return (a collective some rank
    never enters).  On expiry: one line to stderr saying who waited for what, and os._exit(3): the process ends with a fresh non-zero
    exit code (nothing is re-executed, no exception tries to unwind through a blocked C call)."""

    def __init__(self, what, rank, step, seconds=None):
        self.what, self.rank, self.step = what, rank, step
        self.seconds = float(os.environ.get("RSRT_BENCH_WATCHDOG_S", "120")) if seconds is None else float(seconds)
        self.timer = None

    def _expired(self):
        sys.stderr.write("[bench] WATCHDOG: rank %d has waited %.0f s for %s (step %s) — a rank that never entered the collective, or a link that "
                         "never came up; rerun with NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL (or RSRT_COMM_MODE=reduce / --collective torch-nccl) "
                         "to see where.  Exiting with code 3.\n" % (self.rank, self.seconds, self.what, self.step))
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        import threading
        self.timer = threading.Timer(self.seconds, self._expired)
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def fault(name):
    """Fault injection for the watchdog's test: RSRT_BENCH_FAULT=<name> makes this call never return (as a hung collective would)."""
    if os.environ.get("RSRT_BENCH_FAULT") == name:
        while True:
            time.sleep(3600)


def frame_proof(gathered, alone):
    """Is the frame the N ranks rendered and rank 0 gathered the frame ONE GPU renders, bit for bit?  Both [H, W, 4] float32."""
    a = np.ascontiguousarray(gathered, np.float32).view(np.uint32)
    b = np.ascontiguousarray(alone, np.float32).view(np.uint32)
    same = a.shape == b.shape and bool(np.array_equal(a, b))
    out = {"frame_equals_1gpu": same}
    if not same and a.shape == b.shape:
        bad = (a != b).any(axis=-1)
        ys, xs = np.nonzero(bad)
        out["differing_pixels"] = int(bad.sum())
        out["first_differing_pixel_xy"] = [int(xs[0]), int(ys[0])]
    return out


def choose_exchange(candidates, attempt, agree, note=None):
    """The first candidate exchange that is RIGHT on every rank.  attempt(name) -> bool (this rank's verdict; an exception counts as False);
    agree(ok) -> bool: the ranks' verdicts combined (all must say yes; a collective — every rank calls it for every candidate, in the same order).
    Returns the name, or None if none is right."""
    for cand in candidates:
        try:
            ok = bool(attempt(cand))
        except Exception as e:  # noqa: BLE001  (a mode that cannot even run is a mode that is not right)
            ok = False
            if note:
                note(cand, str(e).splitlines()[0][:160] if str(e) else type(e).__name__)
        if agree(ok):
            return cand
    return None


def gather_rank_stats(rank, world, trace_ms, reduce_ms, dist=None):
    """Every rank's kernel / exchange milliseconds per frame, on every rank (torch.distributed all_gather_object; world 1: this rank's)."""
    mine = {"rank": rank, "trace_ms": trace_ms, "reduce_ms": reduce_ms}
    if world == 1 or dist is None:
        return [mine]
    out = [None] * world
    dist.all_gather_object(out, mine)
    return sorted(out, key=lambda r: r["rank"])


# ---------------------------------------------------------------------------------------------- the other BASELINE configs
EXTRA_CONFIGS = [  # (label, scene, width, height, spp, bounces, calls per frame)
    ("config 2: default.toml 1280x720 64 spp 10 bounces", "default", 1280, 720, 64, 10, 1),
    ("config 3: cube.toml 1280x720 128 spp 10 bounces", "cube", 1280, 720, 128, 10, 1),
    ("config 3b: suzanne.toml 1280x720 128 spp 10 bounces (968 triangles: general-BVH walk)", "suzanne", 1280, 720, 128, 10, 1),
    ("config 3c: suzanne grid 4x4 1280x720 32 spp 10 bounces (15,488 triangles: general-BVH walk)", "grid4", 1280, 720, 32, 10, 1),
    ("interactive: house.toml 1920x1080, 16 back-to-back single-sample calls (State::render), 8 bounces", "house", 1920, 1080, 16, 8, 16),
]


def extra_configs(env, device_index, log, frames=3):
    """The BASELINE.json configs the headline line does not carry, each timed for `frames` frames after the timed region (N = 1 only):
    ms per frame, Mrays/s and the oracle-counted roofline fraction (f32 + u32 operations per path counted by liboracle_ops.so on sample 0 of
    a quarter-scale frame — a per-path average — x paths per frame / frame time / 78.64 T/s)."""
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import util
    import rsoderh_raytracing_amd as R
    out = []
    peak = N_SIMD * LANES_PER_SIMD_CYCLE * MAX_CLOCK_HZ / 1e12
    for label, scene, w, h, spp, mb, calls in EXTRA_CONFIGS:
        try:
            if scene.startswith("grid"):
                import make_big_scene
                path = make_big_scene.make(int(scene[4:]))
            else:
                path = scene_path(scene)
            sc = R.Scene.load_toml(path)
            st = R.State.new(sc, env, w, h, device=device_index)
            st.max_bounces = mb

            def frame():
                st.clear()
                if calls == 1:
                    st.render_range(0, spp)
                else:
                    for k in range(calls):
                        st.render_range(k * (spp // calls), spp // calls)

            frame()
            st.synchronize()
            st.stats()
            t = time.perf_counter()
            for _ in range(frames):
                frame()
            st.synchronize()
            dt = (time.perf_counter() - t) / frames
            g = st.stats()
            st.close()
            rays, paths = (g["ext_rays"] + g["shadow_rays"]) / frames, g["paths"] / frames
            _, c = oracle.render(util.oracle_scene(sc), util.oracle_env(env), sc.camera_uniform().view(oracle.CAMERA), w // 4, h // 4, 0, 1, mb,
                                 flags=oracle.FLAG_ANYHIT_SHADOW, n_threads=effective_cores(), fast="ops")
            ops = (c["f32_ops"] + c["int_ops"]) / c["paths"]
            achieved = ops * paths / dt / 1e12
            row = {"workload": label, "ms_per_frame": dt * 1e3, "mrays_s": rays / dt / 1e6, "rays_per_frame": rays, "kernel_ms_per_frame": g["trace_kernel_ms"] / frames,
                   "roofline": {"bound": "valu", "frac": achieved / peak, "achieved": achieved, "peak": peak, "unit": "T lane-operations/s",
                                "ops_per_path": ops, "counted_on": "sample 0 of a %dx%d frame (liboracle_ops.so)" % (w // 4, h // 4)}}
            if calls > 1:
                row["ms_per_call"] = dt * 1e3 / calls
            out.append(row)
            log("extra config: %-90s %8.2f ms/frame %8.0f Mrays/s frac %.3f" % (label, dt * 1e3, rays / dt / 1e6, achieved / peak))
        except Exception as e:  # noqa: BLE001  (a config that cannot run must not cost the headline line)
            out.append({"workload": label, "error": str(e).splitlines()[0][:200]})
            log("extra config %s failed: %s" % (label, e))
    return out


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
    ap.add_argument("--no-pmc", action="store_true", help="do not start rocprofv3 (the committed counters are used if they match the library)")
    ap.add_argument("--no-extra-configs", action="store_true", help="N = 1: do not time the other BASELINE configs after the timed region")
    ap.add_argument("--collective", default="rccl", choices=["rccl", "torch-nccl", "gloo"],
                    help="rccl: rsrt_comm_reduce inside librsrt (the product path); torch-nccl: torch.distributed's reduce; "
                         "gloo: host reduce (rehearsal of N > 1 on one GPU)")
    ap.add_argument("--write-profile", action="store_true", help="store the PMC counters + algorithmic counts under profiles/")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

    # stdout is for the ONE JSON line: libraries that write to fd 1 (RCCL prints a version banner when a communicator
    # comes up) are sent to stderr for the whole run, the line goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL between processes needs dmabuf IPC on this driver
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (before the first HIP call: the pipelined single-sample calls of extra_configs want their lanes on queues of their own)
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

    std_cfg = (args.scene, args.width, args.height, args.bounces, args.spp) == ("house", 1920, 1080, 8, 256)
    # PMC passes first: this process has not touched the GPU yet, the children come and go one at a time
    pmc = None
    if world == 1 and not args.no_pmc:
        pmc = collect_pmc(args, log)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the integrator has no CPU path")
    device_index = local_rank % torch.cuda.device_count()  # == local_rank on a real N-GPU node
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)  # control plane only

    import rsoderh_raytracing_amd as R
    from rsoderh_raytracing_amd import partition, state as S
    scene = R.Scene.load_toml(scene_path(args.scene))
    env = R.Environment.synthetic(2048, 1024)
    W, H, spp = args.width, args.height, args.spp

    state = R.State.new(scene, env, W, H, device=device_index)
    state.max_bounces = args.bounces
    state.set_partition(rank, world, partition.TILE_W, partition.TILE_H)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    state.bind_accumulator(acc.data_ptr(), W, H)
    stream = torch.cuda.Stream()
    build_id = S.build_id()
    log(state.describe() + "; build " + build_id)

    # ---- the exchange step: RCCL inside librsrt; torch's own reduce only if that cannot be set up on every rank
    collective, nccl_group = "none", None
    if world > 1:
        collective = args.collective
        if collective == "rccl":
            # rsrt_comm_init is a collective (ncclCommInitRank): a rank that cannot even load librccl would return early
            # while the others block inside it.  So every rank first asks locally (a dlopen, nothing else) and the answers
            # are combined over gloo; the collective is entered only if ALL ranks can.
            can = torch.tensor([1 if R.State.comm_available() else 0], dtype=torch.int32)
            dist.all_reduce(can, op=dist.ReduceOp.MIN)
            if int(can[0]) == 0:
                log("librccl cannot be loaded on some rank: falling back to torch.distributed's nccl reduce")
                collective = "torch-nccl"
        if collective == "rccl":
            uid, err = [None], ""
            if rank == 0:
                try:
                    uid[0] = R.State.comm_unique_id()
                except R.RsrtError as e:
                    err = str(e)
            dist.broadcast_object_list(uid, src=0)
            ok = torch.tensor([1 if uid[0] is not None else 0], dtype=torch.int32)
            if uid[0] is not None:
                try:
                    with watchdog("rsrt_comm_init (ncclCommInitRank)", rank, "bring-up"):
                        fault("hang_init")
                        state.comm_init(rank, world, uid[0])  # collective: every rank has the id, so every rank calls it
                except R.RsrtError as e:
                    ok[0], err = 0, str(e)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 0:
                log("rsrt_comm_init failed somewhere (%s): falling back to torch.distributed's nccl reduce" % err)
                try:
                    state.comm_destroy()
                except R.RsrtError:
                    pass
                state.set_partition(rank, world, partition.TILE_W, partition.TILE_H)
                collective = "torch-nccl"
        if collective == "torch-nccl":
            # probe it: on a box where the ranks share one GPU (a rehearsal) NCCL refuses the communicator
            ok = torch.tensor([1], dtype=torch.int32)
            try:
                nccl_group = dist.new_group(backend="nccl", device_id=torch.device("cuda", device_index))
                probe = torch.ones(4, device="cuda")
                dist.reduce(probe, dst=0, group=nccl_group)
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                ok[0] = 0
                log("torch nccl reduce unavailable (%s): the accumulators are reduced through gloo on the host" % (str(e).splitlines()[0][:120],))
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 0:
                collective, nccl_group = "gloo", None

    def step():
        acc.zero_()
        state.render_range(0, spp, stream=stream.cuda_stream)
        if world > 1:
            if collective == "rccl":
                state.comm_reduce(0, stream=stream.cuda_stream)  # ONE exchange: each rank's tiles (1/N of the frame) to rank 0 over xGMI
            elif collective == "torch-nccl":
                partition.reduce_accumulators(acc, group=nccl_group)
            else:  # rehearsal: gloo reduces on the host
                host = acc.cpu()
                partition.reduce_accumulators(host)
                if rank == 0:
                    acc.copy_(host)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    exchange_log = []  # what the pre-flight proof found, mode by mode
    with torch.cuda.stream(stream):
        if world > 1:
            # The first exchange ever, under the watchdog, and PROVED before anything is timed: rank 0 renders the whole frame alone once
            # (partition 0 of 1, a second buffer) and every candidate exchange must reproduce it bit for bit — librsrt's gather of compact tile
            # buffers first; if its frame is wrong (it has never run with N > 1 on hardware), librsrt's dense ncclReduce (rsrt_comm_set_mode);
            # then torch's own nccl reduce; then gloo.  All ranks switch together; the timed region runs the first mode that is right.
            alone_frame = None
            if rank == 0:
                alone = R.State.new(scene, env, W, H, device=device_index)
                alone.max_bounces = args.bounces
                alone.render_range(0, spp)
                alone_frame = alone.download()
                alone.close()
            candidates = {"rccl": ["rccl", "rccl-dense-reduce", "torch-nccl", "gloo"], "torch-nccl": ["torch-nccl", "gloo"], "gloo": ["gloo"]}[collective]
            if os.environ.get("RSRT_COMM_MODE") == "reduce" and collective == "rccl":
                candidates = candidates[1:]
            def attempt(cand):
                nonlocal collective, nccl_group
                if cand == "rccl-dense-reduce":
                    state.comm_set_mode(True)
                elif cand == "torch-nccl" and nccl_group is None:
                    nccl_group = dist.new_group(backend="nccl", device_id=torch.device("cuda", device_index))
                collective = "rccl" if cand.startswith("rccl") else cand
                with watchdog("the first exchange (%s)" % cand, rank, "pre-flight frame"):
                    fault("hang_reduce")
                    step()
                    fence()
                if rank != 0:
                    return True
                pf = frame_proof(acc.cpu().numpy(), alone_frame)
                exchange_log.append({"exchange": cand, **pf})
                log("pre-flight: exchange %s reproduces the 1-GPU frame bit for bit: %s" % (cand, pf["frame_equals_1gpu"]))
                return pf["frame_equals_1gpu"]

            def agree(ok):
                t = torch.tensor([1 if ok else 0], dtype=torch.int32)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return int(t[0]) == 1

            def note(cand, msg):
                if rank == 0:
                    exchange_log.append({"exchange": cand, "error": msg})
                log("pre-flight: exchange %s failed: %s" % (cand, msg))

            chosen = choose_exchange(candidates, attempt, agree, note)
            if chosen is None:
                raise SystemExit("no exchange reproduces the 1-GPU frame: %s" % exchange_log)
            exchange_used = chosen
        for _ in range(args.warmup):
            step()
        fence()
        state.stats()  # drop warm-up counters
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0

    s = state.stats()
    tot = torch.tensor([float(s["ext_rays"] + s["shadow_rays"]), float(s["paths"])], dtype=torch.float64)
    tmax = torch.tensor([elapsed, s["trace_kernel_ms"], s["reduce_ms"]], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax[0].item())
    rays_total, paths_total = float(tot[0].item()), float(tot[1].item())
    per_rank = gather_rank_stats(rank, world, s["trace_kernel_ms"] / args.steps, s["reduce_ms"] / args.steps, dist if world > 1 else None)

    result = None
    if rank == 0:
        frame = acc.cpu().numpy()
        assert np.isfinite(frame).all() and np.all(frame[..., 3] == 1.0), "frame incomplete"
        proof = None
        if world > 1:  # the proof again, on the LAST timed frame: against the frame this GPU rendered alone before the timed region
            proof = frame_proof(frame, alone_frame)
            log("N = %d frame equals the 1-GPU frame bit for bit: %s" % (world, proof["frame_equals_1gpu"]))
        value = rays_total / elapsed / 1e6
        launches_trace = args.steps * max(1, (s["launches"] // 2) // args.steps)
        trace_ms_per_launch = s["trace_kernel_ms"] / launches_trace
        cpu_base, per_path, counts = None, None, None
        if world == 1 and not args.no_cpu_baseline:
            cpu_base, per_path, counts = cpu_leg(scene, env, W, H, args.bounces, log)
        committed = None
        if std_cfg and os.path.exists(PMC_FILE):
            with open(PMC_FILE) as f:
                committed = json.load(f)
        if per_path is None and committed is not None:
            per_path = committed.get("algorithmic_bytes_per_path")
        if counts is None and committed is not None and (committed.get("algorithmic_counts_1spp") or {}).get("f32_ops"):
            counts = committed["algorithmic_counts_1spp"]  # (a property of the workload and of the reference's algorithm, not of the kernel)
        if pmc is not None and pmc["build_id"] != build_id:
            log("pmc: the child ran build %s, this process %s: counters dropped" % (pmc["build_id"], build_id))
            pmc = None
        if pmc is None and world == 1 and committed is not None:
            if committed.get("build_id") == build_id:
                pmc = {k: committed[k] for k in ("counters", "resolve", "kernel", "build_id", "launch_ms_under_pmc")}
                pmc["source"] = "committed: profiles/pmc_house_1080p_8b.json (same rsrt_build_id as this library)"
            else:
                log("pmc: profiles/pmc_house_1080p_8b.json was taken on build %s, this library is %s: not attached" % (committed.get("build_id"), build_id))
        paths_per_launch = s["paths"] / launches_trace  # this rank's kernel
        owned_pixels = int(partition.owned_mask(W, H, rank, world).sum())
        roofline = roofline_object(pmc if world == 1 else None, build_id, trace_ms_per_launch, paths_per_launch, per_path, owned_pixels, spp,
                                   counts)
        if world > 1:
            roofline["note"] = "per-kernel counters are collected at N = 1 only; this object carries the launch time and the algorithmic bytes of rank 0's share"
        if args.write_profile and std_cfg and world == 1 and pmc is not None and str(pmc.get("source", "")).startswith("live"):
            os.makedirs(os.path.dirname(PMC_FILE), exist_ok=True)
            with open(PMC_FILE, "w") as f:
                json.dump({"workload": "house.toml 1920x1080 256 spp 8 bounces, one frame per PMC pass", "build_id": build_id, "kernel": pmc["kernel"],
                           "launch_ms_under_pmc": pmc["launch_ms_under_pmc"], "counters": pmc["counters"], "resolve": pmc["resolve"],
                           "algorithmic_bytes_per_path": per_path, "algorithmic_counts_1spp": counts,
                           "passes": PMC_PASSES, "source": "bench.py --write-profile (rocprofv3 --kernel-trace --pmc, separate passes)"}, f, indent=1)
        metric = "Mrays/s, house.toml 1920x1080 256spp 8-bounce" if std_cfg else \
            "Mrays/s, %s %dx%d %dspp %d-bounce" % (os.path.basename(args.scene), W, H, spp, args.bounces)
        result = {"metric": metric, "value": value, "unit": "Mrays/s",
                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                  "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                  "config": {"workload": "%s.toml %dx%d %d spp %d bounces, synthetic 2048x1024 HDRI env" % (os.path.basename(args.scene).replace(".toml", ""), W, H, spp, args.bounces),
                             "parallelism": ("tiles16x16-interleaved-skewed x%d, one exchange per frame via %s" % (world, {"rccl": "librsrt rsrt_comm_reduce (RCCL gather of compact tile buffers)", "torch-nccl": "torch.distributed nccl", "gloo": "gloo on the host (rehearsal)"}[collective]))
                             if world > 1 else "single GPU",
                             "rays_per_frame": rays_total / args.steps, "paths_per_frame": paths_total / args.steps},
                  "ms_per_frame": elapsed / args.steps * 1e3,
                  "kernel_ms_per_frame": {"trace": float(tmax[1].item()) / args.steps, "resolve": s["resolve_kernel_ms"] / args.steps,
                                          "reduce": float(tmax[2].item()) / args.steps},
                  "roofline": roofline, "cpu_baseline": cpu_base}
        if world > 1:
            result["frame_equals_1gpu"] = proof["frame_equals_1gpu"]
            result["multi_gpu"] = {"collective": collective, "exchange": exchange_used, "pre_flight": exchange_log, "collective_note": {"rccl": "librsrt rsrt_comm_reduce: grouped ncclSend / ncclRecv gather of compact tile buffers (RSRT_COMM_MODE=reduce: dense ncclReduce)",
                                                                                  "torch-nccl": "torch.distributed reduce (fallback: librsrt's communicator did not come up)",
                                                                                  "gloo": "host reduce over gloo (rehearsal)"}[collective],
                                   "comm_mode_env": os.environ.get("RSRT_COMM_MODE", ""), "per_rank": per_rank, "proof": proof,
                                   "proof_how": "rank 0 rendered the whole frame alone (partition 0 of 1, second buffer) before the timed region; every candidate exchange "
                                                "had to reproduce it (pre_flight), the last timed frame is compared again; uint32 views"}
        elif std_cfg and not args.no_extra_configs:
            result["extra_configs"] = extra_configs(env, device_index, log)
    state.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if result is not None:
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
