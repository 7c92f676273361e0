"""CPU suite: the C-ABI library loads and exports every symbol include/nblic_amd.h declares;
the host half of the path (serial range-coder stage) is checked against the oracle.  No
compute call that needs a GPU is made here."""
import os
import re

import numpy as np

import inputs


def declared_symbols(header_text):
    names = set()
    for m in re.finditer(r"^[A-Za-z_][\w \*]*?\b(\w+)\s*\(", header_text, re.M):
        if m.group(1) not in ("defined",):
            names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    with open(pkg.INCLUDE) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    names = declared_symbols(text)
    assert {"NBLICcompress", "NBLICdecompress", "QNBLICcompress", "QNBLICdecompress",
            "QNBLICcompressMultiThread", "nblic_amd_encode_batch"} <= names
    assert names == set(pkg.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n
    assert b"NBLIC v0.3" in lib.nblic_amd_version()


def test_limits_match_reference_header(pkg):
    text = open(pkg.INCLUDE).read()
    assert re.search(r"#define\s+NBLIC_MAX_IMG_SIZE\s+100000000", text)
    assert re.search(r"#define\s+NBLIC_MAX_HEIGHT\s+65535", text)


def test_host_range_coder_matches_oracle(pkg, oracle):
    for content, h, w in [("syn1", 64, 64), ("noise", 40, 37), ("const", 1, 1), ("checker", 17, 13)]:
        st = oracle.stages(inputs.make(content, h, w))
        coded = st["prob"].astype(np.uint16) | (st["ev_bin"].astype(np.uint16) << 15)
        assert pkg.range_code(coded) == st["body"]
    # capacity is enforced, not overrun
    st = oracle.stages(inputs.make("noise", 40, 37))
    coded = st["prob"].astype(np.uint16) | (st["ev_bin"].astype(np.uint16) << 15)
    assert pkg.range_code(coded, cap=len(st["body"])) == st["body"]
    assert pkg.range_code(coded, cap=len(st["body"]) - 1) is None
    assert pkg.range_code(np.zeros(0, np.uint16)) == bytes(4)


def test_product_does_not_reference_oracle(pkg):
    # the shipped path must never route through the checker
    root = os.path.dirname(pkg.__file__)
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower().replace("no cpu fallback", ""), os.path.join(dirpath, f)


def test_multi_stream_range_coder_matches_oracle(pkg, oracle):
    """The 8-lane AVX-512 coder (or its scalar stand-in) must give the oracle's bytes for every
    lane, for ragged lengths, fewer / more than eight streams, and must honour capacities."""
    rng = np.random.default_rng(5)
    cases = [("syn1", 64, 64), ("noise", 40, 37), ("const", 1, 1), ("checker", 17, 13), ("syn1", 96, 128), ("ramp", 33, 77),
             ("noise", 64, 64), ("syn1", 100, 30), ("noise", 9, 200), ("checker", 64, 65), ("syn1", 128, 128)]
    coded, want = [], []
    for content, h, w in cases:
        st = oracle.stages(inputs.make(content, h, w))
        coded.append(st["prob"].astype(np.uint16) | (st["ev_bin"].astype(np.uint16) << 15))
        want.append(st["body"])
    for count in (1, 3, 8, 11):
        got, _ = pkg.range_code_multi(coded[:count])
        assert got == want[:count], count
    # random bin streams (all probabilities, long runs of renormalisation)
    rnd = [(rng.integers(1, 4096, n).astype(np.uint16) | (rng.integers(0, 2, n).astype(np.uint16) << 15)) for n in (5000, 1, 0, 777, 4096, 33, 2500, 9)]
    got, _ = pkg.range_code_multi(rnd)
    assert got == [pkg.range_code(r) for r in rnd]
    skew = [np.full(3000, 1 | (1 << 15), np.uint16), np.full(3000, 4095, np.uint16), np.full(2000, 1, np.uint16)]
    got, _ = pkg.range_code_multi(skew)
    assert got == [pkg.range_code(r) for r in skew]
    # capacity: exact fit passes, one byte less fails for that lane only
    caps = [len(b) for b in want[:8]]
    caps[3] -= 1
    got, _ = pkg.range_code_multi(coded[:8], caps)
    assert got[3] is None and [g for i, g in enumerate(got) if i != 3] == [b for i, b in enumerate(want[:8]) if i != 3]


def test_chunked_resumable_coders_match_whole_stream_coding(pkg):
    """The coder threads feed the resumable coders chunk by chunk (scalar for one image, two
    AVX-512 packs in lock-step for 2..16, three for 17..24): any chunking, any mix of stream lengths (lanes drop
    out as their stream ends, whole chunks may be empty for the short ones) must give the bytes
    of coding each stream whole."""
    rng = np.random.default_rng(11)
    lengths = [5000, 1, 0, 777, 4096, 33, 2500, 9, 4999, 5001, 1234, 64, 63, 65, 3000, 17, 128, 5002, 0, 640, 2047, 4097, 52, 53]
    streams = [(rng.integers(1, 4096, n).astype(np.uint16) | (rng.integers(0, 2, n).astype(np.uint16) << 15)) for n in lengths]
    streams[4] = np.full(4096, 1 | (1 << 15), np.uint16)          # long runs of renormalisation
    streams[10] = np.full(1234, 4095, np.uint16)
    whole = [pkg.range_code(s) for s in streams]
    for count in (1, 2, 3, 8, 9, 16, 17, 20, 23, 24):
        for chunk in (1, 7, 64, 1000, 4096, 10000):
            assert pkg.range_code_chunked(streams[:count], chunk) == whole[:count], (count, chunk)
    caps = [len(b) for b in whole]
    caps[5] -= 1
    got = pkg.range_code_chunked(streams, 512, caps)
    assert got[5] is None and [g for i, g in enumerate(got) if i != 5] == [b for i, b in enumerate(whole) if i != 5]
