"""Host range-coder throughput: scalar, N interleaved scalar streams, 8-lane AVX-512 (no GPU needed)."""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("nblic-image-compression_amd")
rng = np.random.default_rng(1)
n = 6_000_000
p1 = np.clip((rng.normal(0.5, 0.35, n) * 4096).astype(np.int64), 1, 4095)
bins = (rng.random(n) < p1 / 4096.0)
base = (p1.astype(np.uint16) | (bins.astype(np.uint16) << 15))
streams = [np.roll(base, 977 * k).copy() for k in range(16)]
def timed(f):
    t = time.perf_counter(); r = f(); return r, time.perf_counter() - t
for _ in range(2):
    r1, t1 = timed(lambda: [pkg.range_code(s) for s in streams[:8]])
    (r8, simd), t8 = timed(lambda: pkg.range_code_multi(streams[:8]))
    (r16, _), t16 = timed(lambda: pkg.range_code_multi(streams))
    r1b = [pkg.range_code(s) for s in streams[8:]]
    print("2 x 8 lanes interleaved: %.0f Mbins/s  equal=%s" % (16 * n / t16 / 1e6, r1 + r1b == r16))
    (r4, _), t4 = timed(lambda: pkg.range_code_multi(streams[:4]))
    print("bytes/bin %.3f  scalar %.0f Mbins/s  avx512-x8 %.0f (per stream %.0f)  4-of-8 lanes %.0f  equal=%s" % (
        len(r1[0]) / n, 8 * n / t1 / 1e6, 8 * n / t8 / 1e6, n / t8 / 1e6, 4 * n / t4 / 1e6, r1 == r8))
