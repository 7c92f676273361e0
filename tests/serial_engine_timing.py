#!/usr/bin/env python3
"""Raster-serial engine (near > 0, efforts 2/3, decoders) against the CPU oracle: ms per call."""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("nblic-image-compression_amd")
from oracle.oracle import Oracle
o = Oracle()
out = {}
def timed(f):
    t = time.perf_counter(); r = f(); return r, time.perf_counter() - t
pkg.compress(pkg.syn1(8, 8, 1), 1, 1)                                      # context + module load
for name, (h, w, near, effort) in {"n2_e1_256": (256, 256, 2, 1), "n0_e2_256": (256, 256, 0, 2), "n0_e3_256": (256, 256, 0, 3),
                                   "n2_e2_512": (512, 512, 2, 2)}.items():
    img = pkg.syn1(h, w, 1)
    (s, rec, _, _), tg = timed(lambda: pkg.compress(img, near, effort))
    (so, *_), tc = timed(lambda: o.encode(img, near, effort))
    d, td = timed(lambda: pkg.decompress(s))
    out[name] = {"ok": bool(s == so and np.array_equal(d[0], rec)), "gpu_enc_ms": round(tg * 1e3, 1), "gpu_dec_ms": round(td * 1e3, 1),
                 "cpu_oracle_enc_ms": round(tc * 1e3, 1), "gpu_us_per_px": round(tg * 1e6 / (h * w), 2)}
print(json.dumps(out))
