"""Regenerates include/rsrt_srgb_table.h (the 255 linear-light thresholds of the sRGB 8-bit encoding)."""
import numpy as np


def inv(s):
    return s / 12.92 if s <= 0.04045 else ((s + 0.055) / 1.055) ** 2.4


if __name__ == "__main__":
    T = [np.float32(inv((k - 0.5) / 255.0)) for k in range(1, 256)]
    for i in range(0, 255, 5):
        print("    " + ", ".join("%.9ef" % float(t) for t in T[i:i + 5]) + ",")
