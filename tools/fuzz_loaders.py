"""Mutation fuzz of the host loaders (csrc/host/scene_load.cpp: TOML subset, OBJ, Radiance .hdr, the --state camera string): damaged inputs must
come back as a clean error (SceneError / ValueError with a message) or load — never crash, hang or read out of bounds.  Meant to run under the
sanitizer build:      bash tools/sanitize_host.sh --fuzz [trials]      (or plainly: python tools/fuzz_loaders.py [trials] [seed])"""
import os, shutil, struct, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import util
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import host

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ASSETS = os.path.join(ROOT, 'tests', 'golden', 'assets')


def mutate(data: bytes) -> bytes:
    b = bytearray(data)
    kind = rng.integers(0, 7)
    n = len(b)
    if kind == 0 and n:  # flip a few bytes
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, n))] = int(rng.integers(0, 256))
    elif kind == 1 and n:  # truncate
        del b[int(rng.integers(0, n)):]
    elif kind == 2 and n:  # drop a span
        i = int(rng.integers(0, n)); del b[i:i + int(rng.integers(1, 40))]
    elif kind == 3 and n:  # duplicate a span
        i = int(rng.integers(0, n)); j = i + int(rng.integers(1, 60)); b[i:i] = b[i:j]
    elif kind == 4:  # insert tokens the grammars care about
        toks = [b'[', b']', b'[[', b']]', b'=', b'"', b'\n', b'#', b',', b'-', b'1e999', b'nan', b'0x', b'/', b'//', b'f 1 2 3 4 5 6 7\n', b'o x\n', b'vn\n',
                b'v 1 2\n', b'f 0/0/0 1/1/1 2/2/2\n', b'f -1//-1 -2//-2 -3//-3\n', b'f 99999999//1 2//2 3//3\n', b'material = "nope"\n', b'999999999999999999999']
        i = int(rng.integers(0, n + 1)); b[i:i] = toks[int(rng.integers(0, len(toks)))]
    elif kind == 5 and n:  # swap two lines
        lines = bytes(b).split(b'\n')
        if len(lines) > 2:
            i, j = rng.integers(0, len(lines), 2); lines[i], lines[j] = lines[j], lines[i]
        b = bytearray(b'\n'.join(lines))
    else:  # replace a number by an extreme one
        import re
        nums = list(re.finditer(rb'-?\d+\.?\d*', bytes(b)))
        if nums:
            m = nums[int(rng.integers(0, len(nums)))]
            b[m.start():m.end()] = [b'-0', b'1e38', b'4294967296', b'-1', b'0.0000000000000000000000000001', b'1e-46', b'65536'][int(rng.integers(0, 7))]
    return bytes(b)


def synth_hdr(w=24, h=10, rle=True):
    """A small Radiance file (new-style RLE scanlines need 8 <= width < 32768)."""
    out = bytearray(b'#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n' % (h, w))
    px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    for y in range(h):
        if not rle:
            out += px[y].tobytes(); continue
        out += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            x = 0
            while x < w:
                run = min(int(rng.integers(1, 20)), w - x)
                if rng.integers(0, 2):
                    out += bytes([128 + run, int(px[y, x, c])])
                else:
                    out += bytes([run]) + px[y, x:x + run, c].tobytes()
                x += run
    return bytes(out)


def main():
    t0 = time.time()
    tomls = {n: open(os.path.join(ASSETS, 'scenes', n), 'rb').read() for n in os.listdir(os.path.join(ASSETS, 'scenes')) if n.endswith('.toml')}
    objs = {n: open(os.path.join(ASSETS, n), 'rb').read() for n in os.listdir(ASSETS) if n.endswith('.obj') and n != 'suzanne.obj'}
    d = tempfile.mkdtemp(prefix='rsrt_fuzz_')
    os.makedirs(os.path.join(d, 'scenes'))
    counts = {'toml ok': 0, 'toml error': 0, 'obj ok': 0, 'obj error': 0, 'hdr ok': 0, 'hdr error': 0, 'state ok': 0, 'state error': 0}
    try:
        for n, data in objs.items():
            open(os.path.join(d, n), 'wb').write(data)
        open(os.path.join(d, 'suzanne.obj'), 'wb').write(open(os.path.join(ASSETS, 'suzanne.obj'), 'rb').read())
        for t in range(trials):
            which = t % 4
            if which == 0:  # a damaged scene file over intact meshes
                name = list(tomls)[int(rng.integers(0, len(tomls)))]
                data = tomls[name]
                for _ in range(int(rng.integers(1, 4))):
                    data = mutate(data)
                p = os.path.join(d, 'scenes', 'fuzz.toml'); open(p, 'wb').write(data)
                try:
                    sc = R.Scene.load_toml(p); counts['toml ok'] += 1
                    assert len(sc.primitives) == len(sc.spheres) + len(sc.plane_descs) + len(sc.triangles)
                except host.SceneError as e:
                    assert str(e), 'empty error message'; counts['toml error'] += 1
            elif which == 1:  # an intact scene file over a damaged mesh
                name = list(objs)[int(rng.integers(0, len(objs)))]
                data = objs[name]
                for _ in range(int(rng.integers(1, 4))):
                    data = mutate(data)
                open(os.path.join(d, name), 'wb').write(data)
                toml = 'house.toml' if name.startswith('house') else ('cube.toml' if name == 'cube.obj' else None)
                if toml is None:
                    body = b'[camera]\npos = [0.0, 1.0, 3.0]\nyaw = 0.0\npitch = 0.0\nfov_y = 100.0\n\n[[materials]]\nname = "m"\ncolor = [1.0, 1.0, 1.0]\nroughness = 0.5\nmetallic = 0.0\n\n[[objects]]\ntype = "Mesh"\npath = "../%s"\nmaterial = "m"\n' % name.encode()
                    p = os.path.join(d, 'scenes', 'one.toml'); open(p, 'wb').write(body)
                else:
                    p = os.path.join(d, 'scenes', toml); open(p, 'wb').write(tomls[toml])
                try:
                    R.Scene.load_toml(p); counts['obj ok'] += 1
                except host.SceneError as e:
                    assert str(e); counts['obj error'] += 1
                open(os.path.join(d, name), 'wb').write(objs[name])
            elif which == 2:  # a damaged Radiance file
                data = synth_hdr(int(rng.integers(8, 40)), int(rng.integers(1, 12)), bool(rng.integers(0, 2)))
                if rng.integers(0, 8):
                    for _ in range(int(rng.integers(1, 4))):
                        data = mutate(data)
                p = os.path.join(d, 'fuzz.hdr'); open(p, 'wb').write(data)
                try:
                    img = host.load_hdr(p); counts['hdr ok'] += 1
                    assert img.ndim == 3 and img.shape[2] == 4
                except ValueError as e:
                    assert str(e); counts['hdr error'] += 1
            else:  # the --state camera string
                good = host.camera_serialize(host.camera_deserialize('AAAAAAAAgD8AAEBAAAAAAAAAAADbD8k/'))
                s = mutate(good.encode()).decode('latin-1') if rng.integers(0, 6) else good
                s = s.replace('\x00', 'A')
                try:
                    host.camera_deserialize(s); counts['state ok'] += 1
                except (ValueError, UnicodeError) as e:
                    counts['state error'] += 1
    finally:
        shutil.rmtree(d, ignore_errors=True)
    print('fuzz_loaders: %d trials (seed %d), %.0f s: %s — no crash, every failure a message' % (trials, seed, time.time() - t0, ', '.join('%s %d' % kv for kv in counts.items())))


if __name__ == '__main__':
    main()
