"""Deterministic synthetic gray planes shared by the fixture generator and the tests.

Pure integer numpy, no RNG state, so the same bytes come out on every machine.
"""
import numpy as np

SMALL_SHAPES = [(1, 1), (1, 7), (7, 1), (2, 2), (3, 5), (5, 3), (17, 13), (64, 64), (256, 2), (2, 256)]
CONTENTS = ["const", "ramp", "checker", "noise", "syn1"]
# (near, effort) classes of SURVEY.md section 8c; effort 0 / near 12 exercise the clamps
PARAM_CLASSES = [(0, 1), (0, 2), (0, 3), (1, 1), (2, 1), (2, 2), (9, 1), (3, 3), (12, 0)]


def syn1(h, w, seed=1):
    """SYN-1: xorshift32 noise on a triangular ramp with a 16-level texture."""
    n = h * w
    xs = np.empty(n, np.uint32)
    s = seed & 0xFFFFFFFF
    # the xorshift chain is serial; do it in python for small frames, in blocks otherwise
    for k in range(n):
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        xs[k] = s
    xs = xs.reshape(h, w)
    i = np.arange(h, dtype=np.int64)[:, None]
    j = np.arange(w, dtype=np.int64)[None, :]
    t = ((i + 2 * j) >> 3) & 511
    base = np.minimum(255, np.abs(t - 256))
    tex = (i ^ j) & 15
    noise = (xs & 7).astype(np.int64) + ((xs >> 3) & 7).astype(np.int64) - 7
    return np.clip(((base * 3) >> 2) + 32 + tex + noise, 0, 255).astype(np.uint8)


def noise(h, w, seed=7):
    idx = np.arange(h * w, dtype=np.uint64) + np.uint64(seed) * np.uint64(1000003)
    v = idx * np.uint64(0x9E3779B97F4A7C15)
    v ^= v >> np.uint64(29)
    v *= np.uint64(0xBF58476D1CE4E5B9)
    v ^= v >> np.uint64(32)
    return (v & np.uint64(255)).astype(np.uint8).reshape(h, w)


def make(content, h, w):
    if content == "const":
        return np.full((h, w), 77, np.uint8)
    if content == "ramp":
        return ((np.arange(h)[:, None] * 3 + np.arange(w)[None, :] * 5) % 256).astype(np.uint8)
    if content == "checker":
        return (((np.arange(h)[:, None] + np.arange(w)[None, :]) & 1) * 255).astype(np.uint8)
    if content == "noise":
        return noise(h, w)
    if content == "syn1":
        return syn1(h, w)
    raise ValueError(content)


def case_id(content, h, w, near, effort):
    return f"{content}_{h}x{w}_n{near}_e{effort}"


KODAK_DIR = "/root/reference/img_kodak"      # read in place, container only; never copied into the repo


def read_gray_bmp(path):
    """8-bit palettised gray BMP as the reference's reader sees it (FileIO.c:170-287): bottom-up
    rows, 4-byte row padding, pixel = palette index (the Kodak files carry an identity palette)."""
    import struct
    raw = open(path, "rb").read()
    off = struct.unpack_from("<I", raw, 10)[0]
    w, h = struct.unpack_from("<ii", raw, 18)
    bpp = struct.unpack_from("<H", raw, 28)[0]
    assert bpp == 8 and raw[:2] == b"BM"
    stride = (w + 3) & ~3
    rows = np.frombuffer(raw, np.uint8, count=stride * abs(h), offset=off).reshape(abs(h), stride)[:, :w]
    return np.ascontiguousarray(rows[::-1] if h > 0 else rows)
