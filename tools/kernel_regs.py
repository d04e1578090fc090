"""Register / scratch use of the wave-pool kernels, read from the code object of the library the build produces (no GPU needed):
python tools/kernel_regs.py [substring of the mangled template arguments, e.g. ILi2ELj1024]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_code_object as t
pat = sys.argv[1] if len(sys.argv) > 1 else ''
for n, m in sorted(t.kernel_metadata().items()):
    if 'pool_kernel' in n and pat in n and (pat or m['private_segment_fixed_size'] or 'ELj1024' in n):
        print(n[len('_Z21rt_render_pool_kernel'):-len('Ev12RenderParams')], m)
