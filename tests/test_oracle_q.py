"""CPU suite: pins the oracle's QNBLIC (effort 0) restatement to the reference -- committed q_*
golden streams, live comparison when oracle/_ref is present, and the closed-form neighbourhood
(what a stateless GPU kernel needs) against the reference's running window."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import inputs


def test_q_golden_streams(oracle, golden):
    manifest, streams = golden
    for (h, w) in inputs.SMALL_SHAPES:
        for content in inputs.CONTENTS:
            img = inputs.make(content, h, w)
            want = streams[f"q_{content}_{h}x{w}"].tobytes()
            assert oracle.qencode(img) == want, (content, h, w)
            dec = oracle.qdecode(want)
            assert dec is not None and np.array_equal(dec, img), (content, h, w)


def test_q_known_answer_1x1(oracle):
    # 1x1 image, pixel 77 -> 62 bytes
    s = oracle.qencode(np.array([[77]], np.uint8))
    assert len(s) == 62 and s[:14].hex() == "51302e320100010062e0ff7f00d2" and s[-4:].hex() == "01000200"


@pytest.mark.parametrize("key", ["syn1s1_512x512_q0"])
def test_q_large_hash(oracle, golden, key):
    from oracle.oracle import syn1
    manifest, _ = golden
    s = oracle.qencode(syn1(512, 512, 1))
    assert len(s) == manifest["large"][key]["len"] and hashlib.sha256(s).hexdigest() == manifest["large"][key]["sha256"]


def test_q_closed_form_neighbourhood_equals_window(oracle):
    lib = oracle.lib
    rng = np.random.default_rng(3)
    for h in range(1, 7):
        for w in range(1, 10):
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
            win = np.zeros(h * w * 11, np.int32)
            lib.orc_q_taps_window(img.ctypes.data_as(C.POINTER(C.c_uint8)), h, w, win.ctypes.data_as(C.POINTER(C.c_int)))
            win = win.reshape(h, w, 11)
            one = np.zeros(11, np.int32)
            for i in range(h):
                for j in range(w):
                    lib.orc_q_taps(img.ctypes.data_as(C.POINTER(C.c_uint8)), w, i, j, one.ctypes.data_as(C.POINTER(C.c_int)))
                    assert np.array_equal(one, win[i, j]), (h, w, i, j, one, win[i, j])


def test_q_vs_live_reference(oracle, reference):
    rng = np.random.default_rng(77)
    for _ in range(30):
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        if rng.random() < 0.6:
            img = (img // 32 + inputs.make("ramp", h, w) // 2 + 40).astype(np.uint8)
        a, b = oracle.qencode(img), reference.qencode(img)
        assert a == b, (h, w)
        assert np.array_equal(reference.qdecode(a), img) and np.array_equal(oracle.qdecode(b), img)
    img = inputs.syn1(200, 300, 4)
    assert oracle.qencode(img) == reference.qencode(img)
