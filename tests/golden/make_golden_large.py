"""Adds the raster-serial-mode goldens (BASELINE configs 4 and 5 and their stand-ins) to
tests/golden/manifest.json["serial"], from the COMPILED reference (oracle/_ref).

Runs only in the build container.  Every case is produced by the unmodified reference
(_ref/libnblic_ref.so), except frames above the reference's own 100,000,000-pixel limit
(NBLIC.h:31), which come from _ref/libnblic_ref_big.so: the same two source files with that one
constant raised on the compiler command line (oracle/Makefile, rule ref_big) -- recorded per case
as "limit_raised": true.  Contains no reference code; lengths and SHA-256 only.

    python tests/golden/make_golden_large.py [--only KEY_SUBSTRING] [--jobs N]
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor, as_completed

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

# (h, w, near, effort, seed)
CASES = [
    (1024, 1024, 0, 2, 1), (1024, 1024, 0, 3, 1), (1024, 1024, 2, 1, 1), (1024, 1024, 2, 2, 1), (1024, 1024, 9, 1, 1),
    (64, 16384, 0, 3, 1), (24, 16384, 2, 2, 1), (16, 16385, 0, 3, 1), (3, 20000, 0, 1, 1), (3, 20000, 2, 2, 1), (3, 20000, 1, 3, 1),
    (8, 16384, 3, 1, 1), (6, 16385, 2, 1, 1),
    (512, 512, 0, 2, 1), (512, 512, 0, 3, 1), (512, 512, 2, 2, 1), (512, 512, 1, 3, 1),
    (1024, 16384, 0, 3, 1),           # a 1/16 band of config 5's frame (same width, same mode)
    (8192, 8192, 2, 2, 1),            # BASELINE config 4
    (16384, 16384, 0, 3, 1),          # BASELINE config 5 (needs the raised limit)
]


def key_of(h, w, near, effort, seed):
    return f"syn1s{seed}_{h}x{w}_n{near}_e{effort}"


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def run_case(case):
    h, w, near, effort, seed = case
    from oracle.oracle import syn1
    big = h * w > 100000000
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libnblic_ref_big.so" if big else "libnblic_ref.so"))
    u8p = C.POINTER(C.c_uint8)
    lib.NBLICcompress.restype = C.c_int
    lib.NBLICcompress.argtypes = [C.c_int, u8p, u8p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    img = syn1(h, w, seed)
    in_sha = sha(img.tobytes())
    out = np.empty(2 * h * w + 4096, np.uint8)
    n, e = C.c_int(near), C.c_int(effort)
    t0 = time.perf_counter()
    ln = lib.NBLICcompress(0, out.ctypes.data_as(u8p), img.ctypes.data_as(u8p), h, w, C.byref(n), C.byref(e))
    dt = time.perf_counter() - t0
    assert ln > 0, case
    s = out[:ln].tobytes()
    return key_of(*case), {"len": ln, "sha256": sha(s), "recon_sha256": sha(img.tobytes()), "input_sha256": in_sha,
                           "limit_raised": bool(big), "ref_seconds": round(dt, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--jobs", type=int, default=4)
    args = ap.parse_args()
    path = os.path.join(HERE, "manifest.json")
    cases = [c for c in CASES if args.only in key_of(*c)]
    with ProcessPoolExecutor(max_workers=args.jobs) as ex:
        futs = [ex.submit(run_case, c) for c in cases]
        for f in as_completed(futs):
            k, v = f.result()
            with open(path) as fh:
                manifest = json.load(fh)
            manifest.setdefault("serial", {})[k] = v
            with open(path, "w") as fh:
                json.dump(manifest, fh, indent=1, sort_keys=True)
            print(k, v["len"], v["sha256"][:16], v["ref_seconds"], "s", flush=True)


if __name__ == "__main__":
    main()
