"""sha256 of every kernel's instruction stream in librsrt.so's gfx950 code object (addresses and encodings stripped), so that a
source clean-up can be checked to leave the product kernels byte for byte what they were:
    python tools/kernel_hash.py [lib.so] > before.txt ; ... edit ... ; python tools/kernel_hash.py > after.txt ; diff before.txt after.txt
"""
import hashlib, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(lib, d):
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
    return co


def kernel_hashes(lib):
    with tempfile.TemporaryDirectory() as d:
        co = code_object(lib, d)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    out, name, lines = {}, None, []
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", ln)
        if m:
            if name:
                out[name] = lines
            name, lines = m.group(1), []
        elif name and ln.strip():
            lines.append(re.sub(r"//.*$", "", ln).strip())  # (the comment carries the address)
    if name:
        out[name] = lines
    return {k: (hashlib.sha256("\n".join(v).encode()).hexdigest()[:16], len(v)) for k, v in out.items()}


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from rsoderh_raytracing_amd import _build
    lib = sys.argv[1] if len(sys.argv) > 1 else _build.HIP_LIB
    for k, (h, n) in sorted(kernel_hashes(lib).items()):
        print("%s %6d %s" % (h, n, k))
