"""include/rsrt_detmath.h against a float64 libm (independent pin of the shared transcendental contract)."""
import numpy as np

import oracle


def _eval(fn, xs, ys=None):
    L = oracle.lib()
    code = {"sin": 0, "cos": 1, "atan2": 2, "asin": 3}[fn]
    return np.array([L.orc_detmath(code, float(x), float(0.0 if ys is None else ys[i])) for i, x in enumerate(xs)], np.float32)


def _max_ulp(out, ref64):
    ulp = np.maximum(np.spacing(np.abs(ref64.astype(np.float32))).astype(np.float64), 1e-45)
    return float((np.abs(out.astype(np.float64) - ref64) / ulp).max())


def test_sin_cos_accuracy():
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-2 * np.pi, 2 * np.pi, 20000), np.linspace(0, 6.2832, 2001)]).astype(np.float32)
    assert _max_ulp(_eval("sin", xs), np.sin(xs.astype(np.float64))) <= 4.0
    assert _max_ulp(_eval("cos", xs), np.cos(xs.astype(np.float64))) <= 4.0


def test_sin_absolute_error_near_zero_crossings():
    xs = (np.float32(np.pi) + np.arange(-50, 50).astype(np.float32) * np.spacing(np.float32(np.pi))).astype(np.float32)
    assert np.abs(_eval("sin", xs) - np.sin(xs.astype(np.float64))).max() < 1e-9


def test_asin_atan2_accuracy():
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-1, 1, 20000), [-1, 1, 0, 0.5, -0.5, 1e-5]]).astype(np.float32)
    assert _max_ulp(_eval("asin", xs), np.arcsin(xs.astype(np.float64))) <= 4.0
    y, x = rng.normal(size=20000).astype(np.float32), rng.normal(size=20000).astype(np.float32)
    assert _max_ulp(_eval("atan2", y, x), np.arctan2(y.astype(np.float64), x.astype(np.float64))) <= 4.0


def test_special_values():
    assert _eval("sin", [0.0])[0] == 0.0 and _eval("cos", [0.0])[0] == 1.0
    assert np.isnan(_eval("asin", [1.0000001])[0]) and np.isnan(_eval("asin", [np.nan])[0])
    assert np.isnan(_eval("sin", [np.nan])[0]) and np.isnan(_eval("cos", [np.inf])[0])
    assert _eval("atan2", [0.0], [0.0])[0] == 0.0
    assert _eval("atan2", [1.0], [0.0])[0] == np.float32(np.pi / 2)
    assert _eval("atan2", [0.0], [-1.0])[0] == np.float32(np.pi)
    assert _eval("asin", [1.0])[0] == np.float32(np.pi / 2)
