"""CPU suite: the product's host-compilable model code (csrc/model.h) and the double-carried
least-squares arithmetic (csrc/lsq_f64.h) that the serial GPU kernels use, compiled into a scalar
harness (tests/host_harness.cpp) and checked against the oracle.  No GPU, no product library paths
beyond the host range coder."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u8p = C.POINTER(C.c_uint8)


@pytest.fixture(scope="module")
def harness():
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libhost_harness.so")
    src = os.path.join(ROOT, "tests", "host_harness.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-o", so, src], check=True)
    lib = C.CDLL(so)
    lib.hh_model_encode.restype = C.c_long
    lib.hh_model_encode.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint16), C.c_long, C.POINTER(C.c_long)]
    lib.hh_check_divide_free.restype = C.c_long
    lib.hh_check_lane_front.restype = C.c_long
    lib.hh_check_symbol_lanes.restype = C.c_long
    lib.hh_check_lane_front.argtypes = [C.c_int, C.c_int]
    return lib


def harness_encode(lib, pkg, img, near, effort):
    h, w = img.shape
    rec = np.empty_like(img)
    coded = np.empty(40 * h * w + 64, np.uint16)
    fb = C.c_long(0)
    n = lib.hh_model_encode(img.ctypes.data_as(u8p), rec.ctypes.data_as(u8p), h, w, near, effort,
                            coded.ctypes.data_as(C.POINTER(C.c_uint16)), coded.size, C.byref(fb))
    k_step = min(max(3 + 2 * near, 3), 16)
    header = b"NBLIC0.3" + bytes([1, h >> 8, h & 255, w >> 8, w & 255, near, k_step, effort])
    return header + pkg.range_code(coded[:n]), rec, fb.value


def test_divide_free_helpers_exhaustive(harness):
    assert harness.hh_check_divide_free() == 0


def test_symbol_bins_lane_layout_matches_the_walk(harness):
    """The decoders compute a symbol's bin probabilities on the lanes (serial_engine.hip decode_symbol): prefix node t on
    lane t, the suffix tree in heap order.  Every (k_step, level pair, symbol): the nodes the reference's walk visits are
    the ones that layout names."""
    assert harness.hh_check_symbol_lanes() == 0


@pytest.mark.parametrize("qnblic", [0, 1])
def test_lane_table_reproduces_the_model(harness, qnblic):
    """csrc/lane_table.h (one model term per lane: the serial kernels' pixel front) walked on the CPU over random,
    extreme, smooth and flat planes of widths 1..64, rows >= 2: predictor, activity, level, context address and
    regressors equal model.h's on the taps the reference's sampling rules deliver."""
    assert harness.hh_check_lane_front(qnblic, 7) == 0


@pytest.mark.parametrize("near,effort", [(0, 1), (0, 2), (0, 3), (2, 1), (2, 2), (3, 3), (9, 1), (1, 3)])
def test_model_headers_and_f64_least_squares_equal_oracle(harness, pkg, oracle, near, effort):
    for (h, w) in [(17, 13), (64, 64), (5, 300)]:
        for content in inputs.CONTENTS:
            img = inputs.make(content, h, w)
            s, rec, _ = harness_encode(harness, pkg, img, near, effort)
            ws, wrec, *_ = oracle.encode(img, near, effort)
            assert s == ws and np.array_equal(rec, wrec), (h, w, content)


def test_f64_least_squares_on_a_photograph_and_noise(harness, pkg, oracle):
    """A Kodak crop (read in place; skipped where the reference tree is absent) and uniform noise at effort 3:
    the exact-range guard may send pixels to the integer redo, the bytes must not change."""
    frames = [inputs.noise(96, 96, 3)]
    kodak = os.path.join(inputs.KODAK_DIR, "05.bmp")
    if os.path.exists(kodak):
        frames.append(np.ascontiguousarray(inputs.read_gray_bmp(kodak)[100:228, 200:360]))
    for img in frames:
        for near, effort in [(0, 3), (2, 2)]:
            s, rec, fallbacks = harness_encode(harness, pkg, img, near, effort)
            ws, wrec, *_ = oracle.encode(img, near, effort)
            assert s == ws and np.array_equal(rec, wrec)
