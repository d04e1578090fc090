"""The general-BVH path (scenes that do NOT run the flat small-scene loop): suzanne (968 triangles, nodes + escape
links in LDS) and the builder-authored suzanne grid (15,488 triangles, everything in global memory), timed by bench.py
with its live PMC passes, plus a bit-exactness check against the oracle at a reduced size.
    python tools/bvh_bench.py [tag]  -> gpurun_out/bvh_bench_<tag>.json"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import make_big_scene
tag = sys.argv[1] if len(sys.argv) > 1 else "run"
grid = make_big_scene.make(4)
rows = []
for label, scene, w, h, spp, mb in [("suzanne 1280x720 128 spp 10 bounces", "suzanne", 1280, 720, 128, 10),
                                    ("suzanne grid 4x4 (15,488 triangles) 1280x720 32 spp 10 bounces", grid, 1280, 720, 32, 10)]:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "3", "--warmup", "1", "--scene", scene,
                        "--width", str(w), "--height", str(h), "--spp", str(spp), "--bounces", str(mb)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        print("%s FAILED: %s" % (label, r.stderr[-500:]), flush=True)
        continue
    b = json.loads(line[-1]); ro = b["roofline"]; hbm, valu, lds = ro.get("hbm") or {}, ro.get("valu") or {}, ro.get("lds") or {}
    row = dict(label=label, ms_per_frame=b["ms_per_frame"], mrays_s=b["value"], rays_per_frame=b["config"]["rays_per_frame"], kernel=ro.get("kernel"), frac=ro.get("frac"),
               issue_frac=valu.get("issue_frac"), lanes=valu.get("lanes_active_per_instruction"), wait_any=valu.get("wait_any_frac_of_wave_cycles"),
               wait_inst_any=valu.get("wait_inst_any_frac_of_wave_cycles"), l2_hit=hbm.get("l2_hit_rate"), fetch_GB=(hbm.get("fetch_bytes_per_launch") or 0) / 1e9,
               write_GB=(hbm.get("write_bytes_per_launch") or 0) / 1e9, fabric_GBps=hbm.get("GB_per_s"), lds_busy=lds.get("array_busy_frac"),
               counters=ro.get("counters"), build_id=ro.get("build_id"))
    rows.append(row)
    print("%-64s %8.2f ms/frame %8.0f Mrays/s  frac %.3f (issue %.2f x lanes %.1f)  wait_any %.2f  L2 hit %.2f  fabric %.0f GB/s  %s" % (
        label, row["ms_per_frame"], row["mrays_s"], row["frac"] or 0, row["issue_frac"] or 0, row["lanes"] or 0, row["wait_any"] or 0, row["l2_hit"] or 0,
        row["fabric_GBps"] or 0, row["kernel"]), flush=True)
# parity at a reduced size: the grid scene against the oracle, bit for bit
import oracle, util
import rsoderh_raytracing_amd as R
env = R.Environment.synthetic(256, 128)
sc = R.Scene.load_toml(grid)
st = R.State.new(sc, env, 240, 135); st.max_bounces = 10
st.render_range(0, 4); img, g = st.download(), st.stats(); st.close()
ref, ost = oracle.render(util.oracle_scene(sc), util.oracle_env(env), sc.camera_uniform().view(oracle.CAMERA), 240, 135, 0, 4, 10, fast=True)
ok = bool(np.array_equal(util.bits(img), util.bits(ref))) and (g["ext_rays"], g["shadow_rays"]) == (ost["ext_rays"], ost["shadow_rays"])
print("grid scene 240x135 x 4 spp vs oracle: bit-exact %s, rays %d" % (ok, g["ext_rays"] + g["shadow_rays"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(rows=rows, grid_bit_exact=ok), open(os.path.join(ROOT, "gpurun_out", "bvh_bench_%s.json" % tag), "w"), indent=1)
