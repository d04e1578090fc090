"""Per-rank kernel time when the BASELINE frame is partitioned over `world` ranks (one GPU plays each rank in turn):
answers whether interleaved 16x16 tiles, which at 1920 / 16 = 120 tiles per row = 0 mod 8 are fixed column stripes per
rank, balance the load.  Writes gpurun_out/part_sweep.json (kept under profiles/)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
import rsoderh_raytracing_amd as R
env = R.Environment.synthetic(2048, 1024)
sc = R.Scene.load_toml(util.scene_path('house'))
rows = []
for (w, h, spp) in ((1920, 1080, 256), (3840, 2160, 32)):
    st = R.State.new(sc, env, w, h); st.max_bounces = 8
    for world in (1, 2, 4, 8):
        ts, rays = [], []
        for rank in range(world):
            st.set_partition(rank, world)
            st.clear(); st.render_range(0, spp); st.synchronize(); st.stats()
            st.clear(); st.render_range(0, spp); st.synchronize(); g = st.stats()
            ts.append(g['trace_kernel_ms']); rays.append(g['ext_rays'] + g['shadow_rays'])
        rows.append(dict(width=w, height=h, spp=spp, world=world, trace_ms_per_rank=ts, rays_per_rank=rays,
                         slowest_over_mean=max(ts) / (sum(ts) / len(ts)), ideal_ms=rows[-world.bit_length() + 1]['trace_ms_per_rank'][0] / world if world > 1 else ts[0]))
        print(f'{w}x{h} {spp} spp, world {world}: trace ms per rank {["%.2f" % t for t in ts]}  slowest/mean {rows[-1]["slowest_over_mean"]:.3f}  '
              f'1-GPU time / world {rows[-1]["ideal_ms"]:.2f}', flush=True)
    st.close()
json.dump(rows, open(os.path.join(ROOT, 'gpurun_out', 'part_sweep.json'), 'w'), indent=1)
