"""The production kernels' resource use, read from the code object the build produced (no GPU needed): the wave-pool
kernels must stay at 4 waves per SIMD (<= 128 VGPRs) WITHOUT scratch memory — a source change that lengthens live ranges
in SHADE shows up here as `.private_segment_fixed_size` > 0 long before anyone times it (it cost the walks 4 % once)."""
import os, re, subprocess, tempfile
import pytest
from rsoderh_raytracing_amd import _build

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_metadata():
    lib = _build.build_hip()
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    out = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in ("vgpr_count", "private_segment_fixed_size", "vgpr_spill_count")}
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="no llvm-readelf")
def test_pool_kernels_keep_four_waves_per_simd_without_scratch():
    md = kernel_metadata()
    # <SV, BLOCK, POOL, TRAV>: the product's choices — flat loop from an LDS image (one workgroup per CU, and the four-workgroup form); the fixed-order walk with its top block in
    # LDS or all in global memory; the near-first tree walk RSRT_FLAG_PRUNE selects
    wanted = ["ILi1ELj1024ELj192ELi2EE", "ILi1ELj256ELj160ELi2EE", "ILi2ELj1024ELj192ELi3EE", "ILi0ELj256ELj160ELi3EE", "ILi2ELj1024ELj192ELi1EE", "ILi0ELj256ELj160ELi1EE",
              "ILi2ELj1024ELj192ELi4EE", "ILi0ELj256ELj160ELi4EE",  # the wide walk, top block in LDS / all global
              "ILi2ELj1024ELj192ELi5EE", "ILi0ELj256ELj160ELi5EE",  # ... for trees deeper than its register stack
              "ILi2ELj1024ELj128ELi6EE", "ILi0ELj256ELj128ELi6EE", "ILi1ELj256ELj128ELi6EE"]  # the cooperative wide walk: node prefix in LDS / all global / a small scene's whole image (A/B)
    for w in wanted:
        names = [n for n in md if n.startswith("_Z21rt_render_pool_kernel" + w)]
        assert len(names) == 1, (w, names)
        m = md[names[0]]
        assert m["vgpr_count"] <= 128, (names[0], m)
        assert m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0, (names[0], m)
