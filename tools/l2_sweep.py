"""A/B of the cache-policy and arena-size knobs of rt_render_pool_kernel on the BASELINE frame.
Each row = one build (compile-time macros -> librsrt_exp_<hash>.so, never the product library) x run-time knobs, timed by
bench.py itself (3 frames) with its live PMC passes, so every row carries ms/frame, L2 hit rate and fabric bytes.
    python tools/l2_sweep.py --build-only     (here: hipcc cross-compiles, the .so files travel with the snapshot)
    python tools/l2_sweep.py                  (on the GPU box) -> gpurun_out/l2_sweep.json"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsoderh_raytracing_amd import _build

ROWS = [  # label, compile-time flags, run-time environment
    ("r01 layout: 14 cold columns, 3 plain dword sample stores", "-DRT_COLD_COMPACT=0 -DRT_SAMPLE_STORE=0", {}),
    ("12 cold columns, 3 plain dword sample stores", "-DRT_SAMPLE_STORE=0", {}),
    ("12 columns, one plain 12-byte store", "-DRT_SAMPLE_STORE=1", {}),
    ("12 columns, one nt 12-byte store (product)", "", {}),
    ("12 columns, one sc1 12-byte store", "-DRT_SAMPLE_STORE=3", {}),
    ("product + nt environment gathers", "-DRT_ENV_NT=1", {}),
    ("product, 3 workgroups per CU", "", {"RSRT_BLOCKS_PER_CU": "3"}),
    ("product, 128 slots per wave", "", {"RSRT_KERNEL": "3"}),
    ("product, 128 slots, 3 workgroups per CU", "", {"RSRT_KERNEL": "3", "RSRT_BLOCKS_PER_CU": "3"}),
    ("sc1 store + nt env, 3 workgroups per CU", "-DRT_SAMPLE_STORE=3 -DRT_ENV_NT=1", {"RSRT_BLOCKS_PER_CU": "3"}),
    ("product + nt on the alias-slot gather only", "-DRT_ENV_NT=2", {}),
    ("sc1 store + nt on the alias-slot gather only", "-DRT_SAMPLE_STORE=3 -DRT_ENV_NT=2", {}),
]
if "--only" in sys.argv:  # e.g. --only 3,10,11
    keep = [int(k) for k in sys.argv[sys.argv.index("--only") + 1].split(",")]
    ROWS = [ROWS[k] for k in keep]


def lib_for(flags):
    env = dict(os.environ)
    if flags:
        os.environ["RSRT_HIPCC_FLAGS"] = flags
    else:
        os.environ.pop("RSRT_HIPCC_FLAGS", None)
    try:
        return _build.build_hip()
    finally:
        os.environ.clear()
        os.environ.update(env)


libs = {flags: lib_for(flags) for _, flags, _ in ROWS}
if "--build-only" in sys.argv:
    for f, l in libs.items():
        print("%-50s %s" % (f or "(product)", os.path.basename(l)))
    sys.exit(0)
out = []
for label, flags, renv in ROWS:
    env = dict(os.environ, RSRT_LIB=libs[flags], **renv)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "3", "--warmup", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        print("%-60s FAILED: %s" % (label, r.stderr[-300:]), flush=True)
        continue
    b = json.loads(line[-1])
    ro = b["roofline"]
    hbm, valu = ro.get("hbm") or {}, ro.get("valu") or {}
    row = dict(label=label, flags=flags, env=renv, ms_per_frame=b["ms_per_frame"], trace_ms=b["kernel_ms_per_frame"]["trace"], frac=ro.get("frac"),
               l2_hit=hbm.get("l2_hit_rate"), fetch_GB=(hbm.get("fetch_bytes_per_launch") or 0) / 1e9, write_GB=(hbm.get("write_bytes_per_launch") or 0) / 1e9,
               wait_any=valu.get("wait_any_frac_of_wave_cycles"), issue_frac=valu.get("issue_frac"), build_id=ro.get("build_id"))
    out.append(row)
    print("%-60s %7.2f ms  trace %7.2f  frac %.3f  L2 hit %.3f  fetch %6.1f GB  write %6.1f GB  wait_any %.3f" % (
        label, row["ms_per_frame"], row["trace_ms"], row["frac"] or 0, row["l2_hit"] or 0, row["fetch_GB"], row["write_GB"], row["wait_any"] or 0), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "l2_sweep.json"), "w"), indent=1)
