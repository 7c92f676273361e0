"""CPU suite: pins the oracle (our CPU restatement) to the reference's outputs.

  * every committed golden stream (generated from the compiled reference by
    tests/golden/make_golden.py) must be reproduced byte for byte, and must decode back;
  * when oracle/_ref is present the oracle is also compared live on further shapes;
  * the staged (key-partitioned) -e1 pipeline must equal the fused engine -- the proof that
    per-key replay is exact.
"""
import hashlib

import numpy as np
import pytest

import inputs


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def test_syn1_python_equals_c():
    from oracle.oracle import syn1
    for (h, w, seed) in [(5, 7, 1), (64, 64, 1), (33, 129, 9)]:
        assert np.array_equal(inputs.syn1(h, w, seed), syn1(h, w, seed))


@pytest.mark.parametrize("content", inputs.CONTENTS)
def test_oracle_reproduces_golden_streams(oracle, golden, content):
    manifest, streams = golden
    for (h, w) in inputs.SMALL_SHAPES:
        img = inputs.make(content, h, w)
        for near, effort in inputs.PARAM_CLASSES:
            cid = inputs.case_id(content, h, w, near, effort)
            want = streams[cid].tobytes()
            got, rec, n_out, e_out, _ = oracle.encode(img, near, effort)
            assert got == want, cid
            m = manifest["small"][cid]
            assert (n_out, e_out) == (m["near_out"], m["effort_out"]), cid
            assert sha(rec.tobytes()) == m["recon_sha256"], cid
            assert int(np.abs(rec.astype(int) - img.astype(int)).max()) <= n_out, cid
            dec = oracle.decode(want)
            assert dec is not None and np.array_equal(dec[0], rec), cid
            assert dec[1:] == (n_out, e_out), cid


def test_known_answer_1x1(oracle):
    # 1x1 image, pixel 77, -n0 -e1 -> 23 bytes
    s = oracle.encode(np.array([[77]], np.uint8), 0, 1)[0]
    assert s.hex() == "4e424c4943302e33010001000100030100000320000000"


@pytest.mark.parametrize("key", ["syn1s1_512x512_n0_e1", "syn1s2_512x512_n0_e1", "syn1s1_512x512_n2_e1",
                                 "syn1s1_256x256_n0_e2", "syn1s1_256x256_n0_e3", "syn1s1_256x256_n2_e2",
                                 "syn1s3_768x512_n0_e1", "syn1s1_1024x1024_n0_e1"])
def test_oracle_matches_large_hashes(oracle, golden, key):
    from oracle.oracle import syn1
    manifest, _ = golden
    m = manifest["large"][key]
    name, dims, n, e = key.split("_")
    seed = int(name[5:]); h, w = map(int, dims.split("x"))
    img = syn1(h, w, seed)
    assert sha(img.tobytes()) == m["input_sha256"]
    s, rec, _, _, _ = oracle.encode(img, int(n[1:]), int(e[1:]))
    assert len(s) == m["len"] and sha(s) == m["sha256"]
    assert sha(rec.tobytes()) == m["recon_sha256"]


def test_survey_appendix_b_constants(golden):
    # the survey's independently measured hashes agree with the regenerated manifest
    manifest, _ = golden
    L = manifest["large"]
    assert L["syn1s1_512x512_n0_e1"]["sha256"].startswith("dadb401b98cd2c62") and L["syn1s1_512x512_n0_e1"]["len"] == 139965
    assert L["syn1s1_4096x4096_n0_e1"]["sha256"].startswith("77d18ede1c1aa384") and L["syn1s1_4096x4096_n0_e1"]["len"] == 8900446
    assert L["syn1s1_512x512_q0"]["sha256"].startswith("50fdb1a0a3cac171")


@pytest.mark.parametrize("shape", [(1, 1), (1, 9), (9, 1), (3, 5), (40, 37), (96, 128), (64, 200)])
def test_staged_equals_fused(oracle, shape):
    h, w = shape
    for content in ("noise", "syn1", "checker", "const"):
        img = inputs.make(content, h, w)
        fused = oracle.encode(img, 0, 1)[0]
        staged, n_ev = oracle.encode_staged(img)
        assert staged == fused, (content, shape)
        st = oracle.stages(img)
        assert len(st["prob"]) == n_ev and int(st["ev_count"].sum()) == n_ev
        assert fused[16:] == st["body"]


def test_limits_and_clamps(oracle):
    img = inputs.make("syn1", 8, 8)
    s9 = oracle.encode(img, 9, 1)[0]
    s12, _, n_out, e_out, _ = oracle.encode(img, 12, 0)
    assert s9 == s12 and (n_out, e_out) == (9, 1)           # near clamps to 9, effort 0 -> 1 (NBLIC.c:768-770)
    assert oracle.decode(b"NOTNBLIC" + bytes(32)) is None
    # pixel-count limit (NBLIC.h:31): 10001 x 10000 is refused before any pixel is touched
    big = np.zeros((1, 1), np.uint8)
    from oracle.oracle import _ptr  # noqa
    import ctypes as C
    out = np.empty(64, np.uint8)
    n, e = C.c_int(0), C.c_int(1)
    assert oracle.lib.orc_nblic_encode(_ptr(out), _ptr(big), 10001, 10000, C.byref(n), C.byref(e), 0, None) == -1


def test_oracle_vs_live_reference(oracle, reference):
    rng = np.random.default_rng(1234)
    for _ in range(25):
        h, w = int(rng.integers(1, 48)), int(rng.integers(1, 48))
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        if rng.random() < 0.5:
            img = (img // 16 + inputs.make("ramp", h, w) // 2).astype(np.uint8)
        near, effort = int(rng.integers(0, 10)), int(rng.integers(1, 4))
        a = oracle.encode(img, near, effort)
        b = reference.encode(img, near, effort)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]), (h, w, near, effort)
        d = reference.decode(a[0])
        assert d is not None and np.array_equal(d[0], a[1])


def test_oracle_on_kodak_matches_reference_and_readme(oracle, golden):
    """BASELINE config 3 content.  The 24 Kodak BMPs are third-party files and are read in place
    (container only); the manifest holds the compiled reference's length + SHA-256 per image, whose
    totals reproduce the reference README's 4.146 bpp (-e1) and 4.227 bpp (-e0)."""
    import os
    manifest, _ = golden
    k = manifest["kodak_e1"]
    assert len(k) == 24
    total_e1 = sum(v["len"] for v in k.values()); total_e0 = sum(v["q_len"] for v in k.values())
    px = sum(v["shape"][0] * v["shape"][1] for v in k.values())
    assert (total_e1, total_e0, px) == (4891174, 4985986, 9437184)
    assert round(8 * total_e1 / px, 3) == 4.146 and round(8 * total_e0 / px, 3) == 4.227   # README.md:256-257
    if not os.path.isdir(inputs.KODAK_DIR):
        pytest.skip("Kodak images are only present in the build container")
    for name in sorted(k):                                                # all 24 images
        img = inputs.read_gray_bmp(os.path.join(inputs.KODAK_DIR, name))
        assert sha(img.tobytes()) == k[name]["input_sha256"]
        s = oracle.encode(img, 0, 1)[0]
        assert (len(s), sha(s)) == (k[name]["len"], k[name]["sha256"]), name
        q = oracle.qencode(img)
        assert (len(q), sha(q)) == (k[name]["q_len"], k[name]["q_sha256"]), name
