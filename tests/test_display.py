"""The 'next' rows of SURVEY.md §8(f): display transform, image output, Radiance .hdr input, --state string.
Checkers here are independent numpy restatements (byte/integer work) — no shared code with the product."""
import base64
import struct

import numpy as np
import pytest
from PIL import Image

import oracle
import util
import rsoderh_raytracing_amd as R
from rsoderh_raytracing_amd import host

TABLE = None


def srgb_thresholds():
    global TABLE
    if TABLE is None:
        def inv(s):
            return s / 12.92 if s <= 0.04045 else ((s + 0.055) / 1.055) ** 2.4
        TABLE = np.array([inv((k - 0.5) / 255.0) for k in range(1, 256)], np.float64).astype(np.float32)
    return TABLE


def display_numpy(sum_rgba, sample_total):
    """hdr.wgsl:3-22 on the f16-rounded mean, then the sRGB 8-bit surface write — plain numpy float32."""
    f = np.float32
    mean = (sum_rgba[..., :3].astype(np.float32) / f(sample_total)).astype(np.float16).astype(np.float32)
    m1 = np.array([[0.59719, 0.07600, 0.02840], [0.35458, 0.90834, 0.13383], [0.04823, 0.01566, 0.83777]], np.float32)  # columns
    m2 = np.array([[1.60475, -0.10208, -0.00327], [-0.53108, 1.10813, -0.07276], [-0.07367, -0.00605, 1.07602]], np.float32)

    def mul(m, v):  # (c0*v0 + c1*v1) + c2*v2, one rounded f32 op at a time
        return (m[0] * v[..., 0:1] + m[1] * v[..., 1:2]) + m[2] * v[..., 2:3]
    v = mul(m1, mean)
    a = v * (v + f(0.0245786)) - f(0.000090537)
    b = v * (f(0.983729) * v + f(0.4329510)) + f(0.238081)
    with np.errstate(all="ignore"):
        r = mul(m2, a / b)
    sdr = np.minimum(np.maximum(r, f(0)), f(1))
    sdr = np.where(np.isnan(r), f(0), sdr)  # clamp as compare-selects: NaN < 0 false, 1 < NaN false -> NaN -> code 0
    neg = (mean < 0).any(axis=-1)
    sdr[neg] = np.float32([1, 0, 1])
    codes = np.searchsorted(srgb_thresholds(), sdr, side="right").astype(np.uint8)
    codes[np.isnan(sdr)] = 0
    return np.concatenate([codes, np.full(codes.shape[:2] + (1,), 255, np.uint8)], axis=-1)


def test_srgb_table_matches_the_standard():
    import re
    text = open(util.ROOT + "/include/rsrt_srgb_table.h").read()
    vals = np.array([float(x) for x in re.findall(r"([0-9.]+e[-+][0-9]+)f", text)], np.float32)
    assert len(vals) == 255 and np.array_equal(vals, srgb_thresholds()) and np.all(np.diff(vals) > 0)
    # code(x) == round(255 * OETF(x)) away from the thresholds
    xs = np.random.default_rng(0).uniform(0, 1, 20000)
    oetf = np.where(xs <= 0.0031308, 12.92 * xs, 1.055 * xs ** (1 / 2.4) - 0.055)
    want = np.rint(255 * oetf)
    got = np.searchsorted(vals, xs.astype(np.float32), side="right")
    assert np.mean(got == want) > 0.999 and np.abs(got - want).max() <= 1


def test_display_transform_matches_numpy_restatement():
    rng = np.random.default_rng(1)
    img = np.zeros((64, 96, 4), np.float32)
    img[..., :3] = (rng.uniform(0, 1, (64, 96, 3)) ** 4 * 40).astype(np.float32)
    img[0, 0, :3] = [-1, 2, 3]          # negative -> magenta
    img[0, 1, :3] = [0, 0, 0]
    img[0, 2, :3] = [1e9, 1e9, 1e9]     # f16 overflow -> inf -> a/b = NaN
    img[0, 3, :3] = [1e-7, 3e-6, 6e-5]  # f16 subnormals
    img[0, 4, :3] = [65519.9 * 7, 65520 * 7, 65504 * 7]  # around the f16 overflow edge (sample_total 7)
    got = host.display_srgb8(img, 7)
    want = display_numpy(img, 7)
    assert np.array_equal(got, want)
    assert list(got[0, 0]) == [255, 0, 255, 255] and list(got[0, 1, :3]) == [0, 0, 0]


def test_png_and_pfm_writers(tmp_path):
    rng = np.random.default_rng(2)
    rgba = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    host.write_png(str(tmp_path / "a.png"), rgba)
    back = np.array(Image.open(tmp_path / "a.png"))
    assert back.shape == rgba.shape and np.array_equal(back, rgba)
    big = rng.integers(0, 256, (300, 400, 4), dtype=np.uint8)  # > 64 KiB: several stored blocks
    host.write_png(str(tmp_path / "b.png"), big)
    assert np.array_equal(np.array(Image.open(tmp_path / "b.png")), big)
    f = rng.normal(size=(5, 7, 4)).astype(np.float32)
    host.write_pfm(str(tmp_path / "a.pfm"), f)
    raw = open(tmp_path / "a.pfm", "rb").read()
    header, rest = raw.split(b"-1.0\n", 1)
    assert header == b"PF\n7 5\n"
    data = np.frombuffer(rest, "<f4").reshape(5, 7, 3)[::-1]
    assert np.array_equal(data, f[..., :3])


def write_rgbe(path, rgbe, rle):
    h, w = rgbe.shape[:2]
    out = bytearray(b"#?RADIANCE\n# test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n" + b"-Y %d +X %d\n" % (h, w))
    for y in range(h):
        if not rle:
            out += rgbe[y].tobytes()
            continue
        out += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            row = rgbe[y, :, c]
            x = 0
            while x < w:
                run = 1
                while x + run < w and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, int(row[x])])
                    x += run
                else:
                    n = 1
                    while x + n < w and n < 128 and not (x + n + 2 < w and row[x + n] == row[x + n + 1] == row[x + n + 2]):
                        n += 1
                    out += bytes([n]) + row[x:x + n].tobytes()
                    x += n
    open(path, "wb").write(out)


@pytest.mark.parametrize("rle", [False, True])
def test_radiance_hdr_reader(tmp_path, rle):
    rng = np.random.default_rng(3)
    rgbe = rng.integers(0, 256, (9, 40, 4), dtype=np.uint8)
    rgbe[:, :10, :] = rgbe[:, :1, :]       # runs
    rgbe[2, 5] = [10, 20, 30, 0]           # e == 0 -> black
    rgbe[3, 6] = [255, 128, 1, 128 + 8]    # scale 1: value = mantissa
    write_rgbe(tmp_path / "t.hdr", rgbe, rle)
    img = host.load_hdr(str(tmp_path / "t.hdr"))
    e = rgbe[..., 3].astype(np.int32)
    want = rgbe[..., :3].astype(np.float32) * np.ldexp(np.float32(1), e - 136)[..., None]
    want[e == 0] = 0
    assert img.shape == (9, 40, 4) and np.array_equal(img[..., :3], want.astype(np.float32)) and np.all(img[..., 3] == 0)
    assert list(img[3, 6, :3]) == [255, 128, 1]
    env = R.Environment.load_hdr(str(tmp_path / "t.hdr"))
    assert env.width == 40 and env.height == 9 and len(env.alias) == 360


def test_radiance_hdr_errors(tmp_path):
    (tmp_path / "x.hdr").write_bytes(b"P6\n1 1\n255\n\0\0\0")
    with pytest.raises(ValueError, match="not a Radiance"):
        host.load_hdr(str(tmp_path / "x.hdr"))
    with pytest.raises(ValueError, match="cannot open"):
        host.load_hdr(str(tmp_path / "missing.hdr"))
    (tmp_path / "y.hdr").write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 2\n\1\2\3\4")
    with pytest.raises(ValueError, match="truncated"):
        host.load_hdr(str(tmp_path / "y.hdr"))


def test_camera_state_string_round_trip():
    d = host.make_camera_desc([0.5, 1.25, -3.0], yaw=0.3, pitch=-0.1, fov_y=1.7453293)
    s = host.camera_serialize(d)
    raw = struct.pack("<6f", 0.5, 1.25, -3.0, np.float32(0.3), np.float32(-0.1), np.float32(1.7453293))
    assert s == base64.b64encode(raw).decode() and len(s) == 32
    back = host.camera_deserialize(s)
    assert back.tobytes() == d.tobytes()
    with pytest.raises(ValueError, match=r"Couldn't deserialize camera: binary data \(21 bytes\) not 24 bytes"):
        host.camera_deserialize(base64.b64encode(b"x" * 21).decode())
    with pytest.raises(ValueError, match="Invalid"):
        host.camera_deserialize("not base64!!")


@pytest.mark.gpu
def test_gpu_display_matches_host_and_numpy(tmp_path):
    sc = R.Scene.load_toml(util.scene_path("house"))
    st = R.State.new(sc, R.Environment.synthetic(256, 128), 160, 90)
    st.max_bounces = 8
    st.render_samples(16)
    sums, dev = st.download(), st.display_srgb8()
    st.close()
    assert np.array_equal(dev, host.display_srgb8(sums, 16))
    assert np.array_equal(dev, display_numpy(sums, 16))
    host.write_png(str(tmp_path / "house.png"), dev)
    assert np.array_equal(np.array(Image.open(tmp_path / "house.png")), dev)
    assert dev[..., :3].std() > 10  # a picture, not a constant


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,ew,eh", [(64, 40, 96, 64), (80, 48, 32, 16)])
def test_developer_views_match_the_shader_restatement(w, h, ew, eh):
    """dev_index 2 and 3 (shader.wgsl:1314-1338; rsrt_debug_view_f16) against oracle.debug_view, the numpy restatement of those lines:
    view 3 shows the environment's texels (zeros outside the map when the frame is larger than it), view 2 adds 0.1 / 20 per draw of the
    alias table onto the previous out_texture through binary16 — here onto zeros and then onto its own result (a second frame with
    another sample_count), with the map larger than the frame (draws that fall outside are dropped) and smaller."""
    sc = R.Scene.load_toml(util.scene_path("default"))
    env = R.Environment.synthetic(ew, eh)
    oenv = util.oracle_env(env)
    st = R.State.new(sc, env, w, h)
    try:
        got3 = st.debug_view(3)
        want3 = oracle.debug_view(3, oenv, w, h, 0)
        assert np.array_equal(got3.view(np.uint16), want3.view(np.uint16))
        got2 = st.debug_view(2, sample_count=0)
        want2 = oracle.debug_view(2, oenv, w, h, 0)
        assert np.array_equal(got2.view(np.uint16), want2.view(np.uint16))
        assert want2[..., :3].max() > 0
        again = st.debug_view(2, out_texture=got2, sample_count=1)  # the next frame brightens what the last one left
        want_again = oracle.debug_view(2, oenv, w, h, 1, out_texture=want2)
        assert np.array_equal(again.view(np.uint16), want_again.view(np.uint16))
        assert float(again[..., :3].astype(np.float32).sum()) > float(got2[..., :3].astype(np.float32).sum())
        with pytest.raises(R.RsrtError, match="dev_index"):
            st.debug_view(1)
    finally:
        st.close()
