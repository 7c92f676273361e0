"""Host range-coder throughput per thread when T threads code at once (no GPU): 16 streams per thread
through the two-pack AVX-512 coder.  Shows what the host's CPU share gives the pipeline's coder stage."""
import importlib, sys, time, os, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("nblic-image-compression_amd")
rng = np.random.default_rng(1)
n = 4_000_000
p1 = np.clip((rng.normal(0.5, 0.35, n) * 4096).astype(np.int64), 1, 4095)
bins = (rng.random(n) < p1 / 4096.0)
base = (p1.astype(np.uint16) | (bins.astype(np.uint16) << 15))
def work(streams, reps, out, i):
    t = time.perf_counter()
    for _ in range(reps): pkg.range_code_multi(streams)
    out[i] = time.perf_counter() - t
for T in (1, 4, 8, 12, 16):
    sets = [[np.roll(base, 977 * (16 * t + k)).copy() for k in range(16)] for t in range(T)]
    out = [0.0] * T
    th = [threading.Thread(target=work, args=(sets[t], 3, out, t)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    wall = time.perf_counter() - t0
    print("threads %2d: %.0f Mbins/s per thread, %.1f Gbins/s together" % (T, 3 * 16 * n / max(out) / 1e6, T * 3 * 16 * n / wall / 1e9))
