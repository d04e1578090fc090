"""python tools/render_png.py <scene> <w> <h> <spp> <bounces> <out.png> — render through the product path and write the display image."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import host
name, w, h, spp, mb, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
sc = R.Scene.load_toml(os.path.join(ROOT, 'tests', 'golden', 'assets', 'scenes', name + '.toml'))
st = R.State.new(sc, R.Environment.synthetic(2048, 1024), w, h); st.max_bounces = mb
st.render_samples(spp)
host.write_png(out, st.display_srgb8())
g = st.stats(); print(name, 'kernel ms', g['kernel_ms'], 'rays', g['ext_rays'] + g['shadow_rays'])
