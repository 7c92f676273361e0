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
